"""Boundary types: soundkit/src/audio_types.rs:3-61 (AudioData) and the frame-header enums it uses."""
import enum
from dataclasses import dataclass

import numpy as np


class EncodingFlag(enum.Enum):
    PCMSigned = 0
    PCMFloat = 1


class Endianness(enum.Enum):
    LittleEndian = 0
    BigEndian = 1


@dataclass
class AudioData:
    """Interleaved PCM bytes + format, the unit crossing the worker's output channel."""
    bits_per_sample: int
    channel_count: int
    sampling_rate: int
    data: np.ndarray  # uint8
    audio_format: EncodingFlag = EncodingFlag.PCMSigned
    endianness: Endianness = Endianness.LittleEndian

    def __post_init__(self):
        if isinstance(self.data, (bytes, bytearray, memoryview)):
            self.data = np.frombuffer(bytes(self.data), np.uint8)
        else:
            self.data = np.ascontiguousarray(self.data).view(np.uint8).ravel()
