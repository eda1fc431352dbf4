"""Hand-off layout constants of the persistent PCG kernels, mirrored from gato_python_amd/csrc/gato_common.h
(pcg_slot_granules, pcg_xslot_granules) for tests and tools."""

MAX_RANKS = 8


def slot_granules(S: int, esz: int) -> int:
    """8-byte granules per workgroup and parity: line 0 = the partial dot, then the first and the last S-block."""
    gpv = esz // 4
    return 16 + ((2 * S * gpv + 15) // 16) * 16


def xslot_granules(S: int, esz: int) -> int:
    """Cross-GPU mirror per parity: one line per rank total, then the left and the right ghost block."""
    return 16 * MAX_RANKS + 2 * (((S * (esz // 4)) + 15) // 16) * 16
