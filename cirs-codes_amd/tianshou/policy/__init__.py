"""Only PPO (the CIRS fork) is on the path; re-exported under the tianshou name for scripts that import it from here."""
from core.policy.ppo import PPOPolicy  # noqa: F401
