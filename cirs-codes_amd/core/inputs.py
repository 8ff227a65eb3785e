"""SparseFeatP + get_dataset_columns (reference core/inputs.py:12-44)."""
from deepctr_torch.inputs import DenseFeat, SparseFeat

DEFAULT_GROUP_NAME = "default_group"


class SparseFeatP(SparseFeat):
    def __new__(cls, name, vocabulary_size, embedding_dim=4, use_hash=False, dtype="int32", embedding_name=None,
                group_name=DEFAULT_GROUP_NAME, padding_idx=None):
        return super().__new__(cls, name, vocabulary_size, embedding_dim, use_hash, dtype, embedding_name, group_name)

    def __init__(self, name, vocabulary_size, embedding_dim=4, use_hash=False, dtype="int32", embedding_name=None,
                 group_name=DEFAULT_GROUP_NAME, padding_idx=None):
        self.padding_idx = padding_idx


def get_dataset_columns(dim_model, envname="VirtualTB-v0", env=None):
    """reference core/inputs.py:24-44: the (user, action, feedback) columns of the state tracker and whether each is embedded."""
    if envname == "VirtualTB-v0":     # 88-d one-hot user, 27-d continuous action, scalar feedback: all dense
        return [DenseFeat("feat_user", 88)], [DenseFeat("feat_item", 27)], [DenseFeat("feat_feedback", 1)], True, True, True
    if envname != "KuaishouEnv-v0":
        raise ValueError(f"unknown env {envname}")
    user_columns = [SparseFeatP("feat_user", env.mat.shape[0], embedding_dim=dim_model)]
    action_columns = [SparseFeatP("feat_item", env.mat.shape[1], embedding_dim=dim_model)]
    feedback_columns = [DenseFeat("feat_feedback", 1)]
    return user_columns, action_columns, feedback_columns, False, False, True


def compute_input_dim(feature_columns, include_sparse=True, include_dense=True, feature_group=False):
    """reference core/user_model.py:538-556"""
    sparse = [f for f in feature_columns if isinstance(f, SparseFeat)]
    dense = [f for f in feature_columns if isinstance(f, DenseFeat)]
    dim = 0
    if include_sparse:
        dim += len(sparse) if feature_group else sum(f.embedding_dim for f in sparse)
    if include_dense:
        dim += sum(f.dimension for f in dense)
    return dim
