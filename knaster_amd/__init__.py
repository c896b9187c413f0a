"""knaster_amd: MI355X-native voice-bank engine for Knaster's UGen hot path.

The product is the C-ABI shared library (include/knaster_hip.h, built from csrc/ by
`python -m knaster_amd.build`); this package is the thin Python plumbing used by the tests,
bench.py and the multi-GPU driver.  Importing the package does not load the library;
the first use of VoiceBank does, and fails loudly if it has not been built.
"""
from . import _lib as lib  # noqa: F401
from .bank import TRIGGER, Stage, VoiceBank, chain_ugen_count, comm_unique_id, shard_voice_range  # noqa: F401

__all__ = ["lib", "Stage", "VoiceBank", "TRIGGER", "chain_ugen_count", "comm_unique_id", "shard_voice_range"]
