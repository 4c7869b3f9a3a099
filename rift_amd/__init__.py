"""rift_amd: MI355X-native RIFT / GRPO policy update."""
import os as _os

# The forward runs its two independent chains on two HIP streams and the trainer keeps the update tail on a third.  HIP maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); in a process that also holds an RCCL communicator (its own streams) four are
# oversubscribed and the chains serialise again: 0.93 ms per step against 0.81 with eight queues (one rank, forced exchanges, MI355X).  Read by the
# HIP runtime when it initialises, hence set at import; an explicit value in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
