"""Host (numpy) statements of the weight-stream layouts documented in include/mi355_llama.h, used to check the
repack kernels bit for bit."""
import numpy as np


def _tiles(N, R, pair):
    rows_per_tile = 16 if pair else 16 * R
    return (N + rows_per_tile - 1) // rows_per_tile


def q4_levels_to_stream(q0: np.ndarray, q1, R: int) -> np.ndarray:
    """levels [N, K] uint8 (0..15) -> Q4 stream bytes.  [tile][unit][r][lane][dword d]; nibble p of dword d holds
    k = 128u + 32g + 8d + j with j = 2 (p & 3) + (p >> 2); lane = 16 g + row."""
    pair = q1 is not None
    N, K = q0.shape
    units = (K + 127) // 128
    tiles = _tiles(N, R, pair)
    mats = [q0, q1] if pair else [q0]
    pad = [np.zeros((tiles * (16 if pair else 16 * R), units * 128), dtype=np.uint32) for _ in mats]
    for m, src in zip(pad, mats):
        m[:N, :K] = src
    words = np.zeros((tiles, units, R, 64, 4), dtype=np.uint32)
    lane = np.arange(64)
    g, row = lane >> 4, lane & 15
    for t in range(tiles):
        for u in range(units):
            for r in range(R):
                src = pad[r] if pair else pad[0]
                n = (t * 16 + row) if pair else ((t * R + r) * 16 + row)
                for d in range(4):
                    w = np.zeros(64, dtype=np.uint32)
                    for p in range(8):
                        j = 2 * (p & 3) + (p >> 2)
                        k = 128 * u + 32 * g + 8 * d + j
                        w |= src[n, k] << np.uint32(4 * p)
                    words[t, u, r, :, d] = w
    return words.reshape(-1).view(np.uint8)


def q4_stream_to_levels(stream: np.ndarray, N: int, K: int, R: int, pair: bool) -> np.ndarray:
    units = (K + 127) // 128
    tiles = _tiles(N, R, pair)
    words = np.ascontiguousarray(stream).view(np.uint32).reshape(tiles, units, R, 64, 4)
    out = np.zeros((2 if pair else 1, tiles * (16 if pair else 16 * R), units * 128), dtype=np.uint8)
    lane = np.arange(64)
    g, row = lane >> 4, lane & 15
    for t in range(tiles):
        for u in range(units):
            for r in range(R):
                n = (t * 16 + row) if pair else ((t * R + r) * 16 + row)
                for d in range(4):
                    for p in range(8):
                        j = 2 * (p & 3) + (p >> 2)
                        k = 128 * u + 32 * g + 8 * d + j
                        out[r if pair else 0, n, k] = (words[t, u, r, :, d] >> np.uint32(4 * p)) & 0xF
    return out[:, :N, :K]


def bf16_bits_to_stream(w_bits: np.ndarray, R: int) -> np.ndarray:
    """bf16 bit patterns [N, K] uint16 -> BF16 stream: [tile][unit][r][piece d][lane][8 x bf16],
    piece d of lane (g, row) holds k = 128u + 32g + 8d + 0..7."""
    N, K = w_bits.shape
    units = (K + 127) // 128
    tiles = _tiles(N, R, False)
    pad = np.zeros((tiles * 16 * R, units * 128), dtype=np.uint16)
    pad[:N, :K] = w_bits
    out = np.zeros((tiles, units, R, 4, 64, 8), dtype=np.uint16)
    lane = np.arange(64)
    g, row = lane >> 4, lane & 15
    for t in range(tiles):
        for u in range(units):
            for r in range(R):
                n = (t * R + r) * 16 + row
                for d in range(4):
                    for j in range(8):
                        out[t, u, r, d, :, j] = pad[n, 128 * u + 32 * g + 8 * d + j]
    return out.reshape(-1).view(np.uint8)


def i8_to_stream(cb: np.ndarray, R: int) -> np.ndarray:
    """int8 [N, K] -> I8 stream: [tile][unit][r][piece e][lane][16 x int8], k = 128u + 64e + 16g + j."""
    N, K = cb.shape
    units = (K + 127) // 128
    tiles = _tiles(N, R, False)
    pad = np.zeros((tiles * 16 * R, units * 128), dtype=np.int8)
    pad[:N, :K] = cb
    out = np.zeros((tiles, units, R, 2, 64, 16), dtype=np.int8)
    lane = np.arange(64)
    g, row = lane >> 4, lane & 15
    for t in range(tiles):
        for u in range(units):
            for r in range(R):
                n = (t * R + r) * 16 + row
                for e in range(2):
                    for j in range(16):
                        out[t, u, r, e, :, j] = pad[n, 128 * u + 64 * e + 16 * g + j]
    return out.reshape(-1).view(np.uint8)
