"""MI355X-native SlotDiffusion hot path."""
import os as _os

# HIP-graph execution: the captured train step has parallel branches (weight gradients on side
# streams); this runtime maps graph branches onto a pool of hardware queues, and with 2 queues the
# cross-queue signalling costs least (same-box A/B of the graphed step: 31.5 / 32.1 ms against
# 32.7 / 32.8 ms with the default pool, 34.8 with 3).  Read by the HIP runtime when it initialises,
# i.e. at the first device call -- a value set by the user wins.
_os.environ.setdefault('DEBUG_HIP_FORCE_GRAPH_QUEUES', '2')
