"""onebit_amd -- MI355X-native implementation of OneBit's 1-bit linear inference path.

Public surface mirrors the reference's for this path:
  BitLinearInf / OneBitLinear   drop-in nn.Module (reference: models/bitnet.py:71-122)
  fp16_to_int8 / pack_signs     sign packer (reference: scripts/convert_llama_to_infer_ckpt.py:7-15)
  int8_to_fp16                  unpacker (reference: models/bitnet.py:98-110)
"""
from .bitnet import BitLinearInf, OneBitLinear, fp16_to_int8, int8_to_fp16, pack_signs  # noqa: F401

__version__ = "0.1.0"
