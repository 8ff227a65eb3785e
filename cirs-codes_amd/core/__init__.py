"""Mirror of the reference's `core` package for the hot path (module and class names of SURVEY 8(b)).  Importing it also makes
`from torch.utils.tensorboard import SummaryWriter` (CIRS-RL-kuaishou.py:18, after its first `core.*` import) resolvable on an image
without the tensorboard package (cirs_hip.compat)."""
from cirs_hip import compat as _compat

_compat.ensure_tensorboard()
