from tianshou.trainer.utils import gather_info, test_episode  # noqa: F401
