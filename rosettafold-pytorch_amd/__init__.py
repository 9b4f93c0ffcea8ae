"""rosettafold-pytorch_amd: MI355X-native RoseTTAFold forward path (hand-written HIP kernels behind the
reference's nn.Module call surface).  Importing this package loads librfmi.so and raises if it is missing:
there is no CPU / PyTorch fallback."""
from . import _lib  # noqa: F401  (fails loudly without the HIP library)
from . import ops  # noqa: F401
from .model import *  # noqa: F401,F403
from .model import set_compute_dtype, RT  # noqa: F401
from .structure import *  # noqa: F401,F403
from .structure import flat_state  # noqa: F401
from .model import invalidate_weight_caches  # noqa: F401
from . import weights  # noqa: F401
from . import featurize  # noqa: F401
from . import shard  # noqa: F401  (batch shards / pair-track row blocks over torch.distributed ranks)
from .graph import GraphedForward  # noqa: F401
from .structure import check_edge_capacity  # noqa: F401
from .weights import export_hidden_lists, load_reference_weights, save_checkpoint, load_checkpoint  # noqa: F401
