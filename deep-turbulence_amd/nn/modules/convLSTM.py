"""Convolutional LSTM on the HIP path.  API mirror of the reference's nn/modules/convLSTM.py
(ConvLSTMCell :16-104, ResidLSTMBlock :107-152)."""
import torch
import torch.nn as nn

import tmg_hip as H
import tmg_ops as ops


class ConvLSTMCell(nn.Module):
    """gates = conv3x3(cat(input, h)) split as i, f, o, g (reference :72-83).  The 4R-wide gate conv
    is the largest contraction of the model and runs on the fp32 matrix cores; the gate
    non-linearities and state update are one pointwise kernel."""

    def __init__(self, input_dim, hidden_dim, kernel_size, bias=True):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.kernel_size = kernel_size
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.bias = bias
        self.conv = nn.Conv2d(in_channels=input_dim + hidden_dim, out_channels=4 * hidden_dim, kernel_size=kernel_size,
                              padding=self.padding, bias=bias)
        self.h0_in = None
        self.c0_in = None

    def run(self, inputs, state):
        """inputs: list of NHWC tensors (concatenated input); state: (h, c) NHWC or None (zeros, reference :87-104)."""
        if state is None:
            t = inputs[0]
            h_cur = torch.zeros(t.shape[:3] + (self.hidden_dim,), device=t.device, dtype=t.dtype)
            c_cur = None
        else:
            h_cur, c_cur = state
        if self.conv.bias is None or tuple(self.kernel_size) != (3, 3):
            gates = ops.conv(list(inputs) + [h_cur], self.conv.weight, self.conv.bias)
            return ops.LSTMPointwiseFn.apply(gates, c_cur)
        return ops.ConvLSTMCellFn.apply(self.conv.weight, self.conv.bias, h_cur, c_cur, *inputs)

    def forward(self, input_tensor, cur_state):
        st = None if cur_state is None else (H.nhwc(cur_state[0]), H.nhwc(cur_state[1]))
        h, c = self.run([H.nhwc(input_tensor)], st)
        return H.nchw(h), H.nchw(c)

    def init_hidden(self, input_tensor):
        dims = list(input_tensor.size())
        dims[1] = self.hidden_dim
        z = torch.zeros(dims, device=input_tensor.device, dtype=input_tensor.dtype)
        return z, z.clone()


class ResidLSTMBlock(nn.Module):
    """ConvLSTM cell followed by relu(conv3x3(cat(input, h_next)) + b) (reference :137-152)."""

    def __init__(self, input_dim, hidden_dim, output_dim, kernel_size, bias=True):
        super().__init__()
        self.convLSTM = ConvLSTMCell(input_dim, hidden_dim, kernel_size, bias)
        self.out_seq = nn.Sequential()
        self.out_seq.add_module('LSTM_out_conv', nn.Conv2d(in_channels=hidden_dim + input_dim, out_channels=output_dim,
                                                           kernel_size=3, padding=1, stride=1))

    def run(self, inputs, state, out_grad_premasked=False):
        """out_grad_premasked: see tmg_ops.conv (the LSTM coupling layer's tail masks the gradient it sends back by [out > 0])."""
        h_next, c_next = self.convLSTM.run(inputs, state)
        oc = self.out_seq.LSTM_out_conv
        out = ops.conv(list(inputs) + [h_next], oc.weight, oc.bias, relu_out=True, _grad_premasked=out_grad_premasked)
        return out, h_next, c_next

    def forward(self, input_tensor, cur_state=None):
        st = None if cur_state is None else (H.nhwc(cur_state[0]), H.nhwc(cur_state[1]))
        out, h, c = self.run([H.nhwc(input_tensor)], st)
        return H.nchw(out), H.nchw(h), H.nchw(c)
