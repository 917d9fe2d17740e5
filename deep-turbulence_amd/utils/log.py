"""Console + file logger with the interface `main.py:36-38` and the loaders / trainer use (reference utils/log.py:29-117):
log / info / success / warning / error, optionally appended to `<args.run_dir>/<file_base><timestamp>.dat`."""
import os
import sys
from time import localtime, strftime

_COLOURS = {"Info": "\033[94m", "Success": "\033[92m", "Warning": "\033[93m", "Error": "\033[91m"}
_STAMP = strftime("%Y-%b-%d_%H_%M_%S", localtime())


class Log(object):
    def __init__(self, args=None, file_base='output', record=True):
        self.record = bool(record) and args is not None
        self.log_file = os.path.join(args.run_dir, file_base + _STAMP + '.dat') if args is not None else None

    def _emit(self, kind, msg, rec=True):
        line = "{}[{}]: {}".format(strftime("[%H:%M:%S]", localtime()), kind, msg)
        colour = _COLOURS.get(kind)
        print(colour + line + "\033[0m" if colour else line)
        if self.record and rec:
            with open(self.log_file, 'a') as f:
                f.write(line + '\n')

    def log(self, str0, rec=True):
        self._emit("Output", str0, rec)

    def info(self, str0):
        self._emit("Info", str0)

    def success(self, str0):
        self._emit("Success", str0)

    def warning(self, str0):
        self._emit("Warning", str0)

    def error(self, str0):
        self._emit("Error", str0)

    def print_progress(self, iteration, total, prefix='', suffix='', decimals=1, bar_length=50):
        frac = iteration / float(total)
        n = int(round(bar_length * frac))
        sys.stdout.write('\r%s |%s| %.*f%% %s' % (prefix, '#' * n + '-' * (bar_length - n), decimals, 100 * frac, suffix))
        if iteration == total:
            sys.stdout.write('\n')
