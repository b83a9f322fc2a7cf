"""Drop-in for the reference module plen_ros_helpers.td3 (plen_ros/src/plen_ros_helpers/td3.py)."""
from plen_ml_walk_amd.td3 import Actor, Critic, ReplayBuffer, TD3Agent, evaluate_policy  # noqa: F401
