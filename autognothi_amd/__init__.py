

import os as _os

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues per priority (default 4).  The two-stream training epoch
# (scripts/common.TrainPartition) and the dW side streams of the training step want their streams on queues of their own: with 8 the
# second stream pays in every sequence measured (vanilla ViT-base +15 %, froyo +8 %, duo BERT +9 %), with 4 it depends on how many
# streams the process happened to create before (and the epoch then falls back to one stream: scripts/common.pipelined_targets).
# Read by the HIP runtime when it initialises, i.e. only effective when this package is imported before the first GPU call; a value
# the user has set is left alone.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
