"""Names the reference exports from core/user_model.py that the hot path uses."""
from core.inputs import compute_input_dim  # noqa: F401
from deepctr_torch.inputs import build_input_features  # noqa: F401
