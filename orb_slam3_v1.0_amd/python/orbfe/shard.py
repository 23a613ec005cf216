"""Frame sharding + result packing for the batched many-frame mode (BASELINE config 4).

Frames are independent units: rank r of N owns a contiguous block (weak scaling in bench.py: every rank
brings its own block).  The only collective is the gather of the padded per-frame results."""
import numpy as np

SLOT_BYTES = 24 + 32 + 4  # keypoint + descriptor + match index


def shard_range(n_frames, rank, world):
    """Contiguous block of frames owned by `rank` (first ranks take the remainder)."""
    per = (n_frames + world - 1) // world
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def pack_results(kp, desc, match):
    """[B, cap] keypoints (24 B), [B, cap, 32] descriptors, [B, cap] int32 matches -> [B, cap, 60] u8."""
    B, cap = kp.shape
    out = np.zeros((B, cap, SLOT_BYTES), np.uint8)
    out[:, :, :24] = kp.view(np.uint8).reshape(B, cap, 24)
    out[:, :, 24:56] = desc
    out[:, :, 56:] = np.ascontiguousarray(match, np.int32).view(np.uint8).reshape(B, cap, 4)
    return out


def unpack_results(packed, kp_dtype):
    B, cap, _ = packed.shape
    kp = np.ascontiguousarray(packed[:, :, :24]).view(kp_dtype).reshape(B, cap)
    desc = np.ascontiguousarray(packed[:, :, 24:56])
    match = np.ascontiguousarray(packed[:, :, 56:]).view(np.int32).reshape(B, cap)
    return kp, desc, match
