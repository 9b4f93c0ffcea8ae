"""rosettafold-pytorch_amd: MI355X-native RoseTTAFold forward path (HIP kernels behind the reference's
nn.Module call surface).  Importing this package loads librfmi.so; it raises if the library is missing."""
from . import _lib  # noqa: F401  (fails loudly without the HIP library)
from . import ops  # noqa: F401
