"""`pc` of the MI355X path: physicsConstrained lives here (fused HIP kernel); the reference's `pc.grad1Filter` /
`pc.grad2Filter` keep resolving from its own directory when that is on sys.path behind this one."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
