"""`utils` of the MI355X path.  This directory provides dataLoader, utils, log and parallel; when the reference's
`tmglow/` directory is ALSO on sys.path (behind this one), the portion below lets the modules that exist only there
(e.g. `utils.viz`, plotting - out of scope here) keep resolving instead of being shadowed by this package."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
