"""zinc_amd -- MI355X-native Zip PCS commit/open path for NethermindEth/zinc.

Layout
  csrc/     hand-written gfx950 HIP kernels + the C ABI (include/zip_hip.h) -> lib/libzip_hip.so
  host/     C++ mirror of the reference's host-side interface for this path
            (RaaCode, MultilinearZip::{setup,commit,open}, PcsTranscript, KeccakTranscript) -> lib/libzinc_zip.so
  cabi.py   ctypes binding of the C ABI (numpy / torch device pointers in, bytes out)
  pcs.py    ctypes binding of the C++ host mirror
  dist.py   one-process-per-GPU row sharding over torch.distributed (RCCL)

Nothing here computes on the CPU: without libzip_hip.so and a gfx950 device every
entry point raises.
"""
__version__ = "0.1.0"
