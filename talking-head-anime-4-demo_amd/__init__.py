"""tha4_amd - MI355X-native (gfx950) hot path of pkhungurn/talking-head-anime-4-demo.

The directory is named ``talking-head-anime-4-demo_amd`` (not importable with a plain ``import``
statement because of the hyphens); ``import tha4_amd`` at the repo root aliases it.

Layout (only what the per-frame poser path needs):
  csrc/            HIP kernels + the C-ABI shared library (include/tha4_hip.h)
  _capi.py         ctypes binding of that ABI (no fallback: raises if the .so is missing)
  poser/           mirror of the reference's ``tha4.poser`` interface (Poser ABC, pose-parameter
                   metadata, modes/mode_14.create_poser) on top of the C ABI
  weights.py       reference ``.pt`` state_dict ingest
  sharding.py      frame-parallel multi-GPU driver (one process per GPU, RCCL gather of finished frames)
"""
__all__ = ["poser", "weights", "sharding"]
