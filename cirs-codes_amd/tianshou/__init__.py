"""The subset of Tianshou 0.4.2's surface that CIRS's hot path touches, backed by the HIP engines in `cirs_hip`
(reference: the vendored tianshou/ tree).  Everything else of Tianshou is out of scope."""
__version__ = "0.4.2-cirs-hip"
