"""SparseFeat / DenseFeat descriptors (reference DeepCTR-Torch/deepctr_torch/inputs.py:20-38,80-87) and
build_input_features (:99-123).  Pure data: no kernels involved."""
from collections import OrderedDict, namedtuple

DEFAULT_GROUP_NAME = "default_group"


class SparseFeat(namedtuple("SparseFeat", ["name", "vocabulary_size", "embedding_dim", "use_hash", "dtype", "embedding_name", "group_name"])):
    __slots__ = ()

    def __new__(cls, name, vocabulary_size, embedding_dim=4, use_hash=False, dtype="int32", embedding_name=None, group_name=DEFAULT_GROUP_NAME):
        if embedding_name is None:
            embedding_name = name
        if embedding_dim == "auto":
            embedding_dim = 6 * int(pow(vocabulary_size, 0.25))
        return super().__new__(cls, name, vocabulary_size, embedding_dim, use_hash, dtype, embedding_name, group_name)

    def __hash__(self):
        return self.name.__hash__()


class DenseFeat(namedtuple("DenseFeat", ["name", "dimension", "dtype"])):
    __slots__ = ()

    def __new__(cls, name, dimension=1, dtype="float32"):
        return super().__new__(cls, name, dimension, dtype)

    def __hash__(self):
        return self.name.__hash__()


def build_input_features(feature_columns):
    features, start = OrderedDict(), 0
    for feat in feature_columns:
        if feat.name in features:
            continue
        width = 1 if isinstance(feat, SparseFeat) else feat.dimension
        features[feat.name] = (start, start + width)
        start += width
    return features
