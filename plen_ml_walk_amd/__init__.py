"""plen_ml_walk_amd: MI355X-native vectorised PLEN walking environment + TD3."""
import os

__version__ = "0.2.0"

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue serialise.  The
# pipelined trainer (three streams that must overlap) and the sub-batch env mode (two) sit next to torch's own side streams: with 4 queues
# the three-stream loop lands in a 0.95-1.9 ms/step regime depending on which streams happened to share a queue; with 8 it is a steady
# 0.73 ms on its own (scripts/gpu_pipeline_probe.py) but 0.95 ms after bench.py's env legs have used up stream slots; from 12 on it is
# 0.73 ms there too.  16 it is.  Only effective if set before the first HIP call of the process, hence here.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
