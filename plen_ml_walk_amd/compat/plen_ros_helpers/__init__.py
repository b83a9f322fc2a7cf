"""Import shim for the reference package plen_ros_helpers (plen_ros/src/plen_ros_helpers)."""
