"""spliser_amd: the host side of libspliser_hip.so (see README.md)."""
import os

# (before anything starts the HIP runtime, torch included: spl_create's comment in csrc/spl_capi.cpp says why)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
