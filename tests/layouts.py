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


def u8_to_stream(q: np.ndarray, R: int) -> np.ndarray:
    """uint8 levels [R][N, K] of 8-bit ColBlock linears (R = 2: the c_fc1 / c_fc2 pair) -> the stream of mi355_fused_step's weight_fmt 6:
    [tile][unit][r][piece e][lane = 16 g + row][16 B], byte b = column 128 u + 32 g + 16 e + 8 (b >> 3) + (0 4 1 5 2 6 3 7)[b & 7] — the
    two pieces of a unit give `v & 0x0F0F0F0F` (low nibbles) and `(v >> 4) & 0x0F0F0F0F` (high nibbles) as the 32 A bytes of lane (g, row)
    in the octet order of the limb planes (f8_planes below): y = sum_k (l_k + 16 h_k) x_k from two scaled MFMAs against ONE B operand."""
    q = np.asarray(q, dtype=np.uint8).reshape(R, *q.shape[-2:])
    _, N, K = q.shape
    assert N % 16 == 0 and K % 128 == 0
    perm = (0, 4, 1, 5, 2, 6, 3, 7)
    out = np.zeros((N // 16, K // 128, R, 2, 64, 16), dtype=np.uint8)
    lane = np.arange(64)
    g, row = lane >> 4, lane & 15
    for t in range(N // 16):
        for u in range(K // 128):
            for r in range(R):
                for e in range(2):
                    for b in range(16):
                        out[t, u, r, e, :, b] = q[r, 16 * t + row, 128 * u + 32 * g + 16 * e + 8 * (b >> 3) + perm[b & 7]]
    return out.reshape(-1)


# ---- hand-off format of the persistent int4 step with fp8 operands (csrc/fused_step_ring.hip FMT 3, mi355_fused_step_args.weight_fmt = 3)
def e4m3_decode(code: np.ndarray) -> np.ndarray:
    """OCP E4M3 (bias 7, subnormals m * 2^-9, no infinities; 0x7F / 0xFF = NaN) -> float64."""
    code = np.asarray(code, dtype=np.uint8)
    s = (code >> 7).astype(np.int64)
    e = ((code >> 3) & 15).astype(np.int64)
    m = (code & 7).astype(np.float64)
    mag = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * 2.0 ** (e - 7.0))
    mag = np.where((code & 0x7F) == 0x7F, np.nan, mag)
    return np.where(s == 1, -mag, mag)


_E4M3_POS = e4m3_decode(np.arange(0, 0x7F, dtype=np.uint8))  # ascending magnitudes of codes 0x00 .. 0x7E


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float -> OCP E4M3 code, clamped to +-448, round to nearest, ties to even (what f8_limbs' v_med3_f32 + v_cvt_pk_fp8_f32 do;
    measured by scripts/micro/mx_fp8.hip)."""
    x = np.asarray(x, dtype=np.float64)
    a = np.minimum(np.abs(x), 448.0)
    hi = np.searchsorted(_E4M3_POS, a, side="left").clip(0, len(_E4M3_POS) - 1)
    lo = (hi - 1).clip(0)
    dlo, dhi = np.abs(a - _E4M3_POS[lo]), np.abs(_E4M3_POS[hi] - a)
    pick_hi = (dhi < dlo) | ((dhi == dlo) & (hi % 2 == 0))  # the code's low bit is the mantissa's: even code = even mantissa
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    return np.where(np.signbit(x), code | 0x80, code).astype(np.uint8)


def f8_limbs(x: np.ndarray) -> np.ndarray:
    """x -> [3, ...] E4M3 codes with x ~ l0 + l1 / 16 + l2 / 256 (residual splitting, every difference exact)."""
    x = np.asarray(x, dtype=np.float64)
    l0 = e4m3_encode(x)
    r = x - e4m3_decode(l0)
    l1 = e4m3_encode(r * 16.0)
    r = r - e4m3_decode(l1) / 16.0
    l2 = e4m3_encode(r * 256.0)
    return np.stack([l0, l1, l2])


def f8_planes(x: np.ndarray, tag: int = 0):
    """A K-vector -> (granules uint64 [K / 2], limb planes uint8 [3, K]) as the publishers write them and the gatherers stage them:
    granule 4 O + j of octet O carries the values at offsets (j, j + 4) as bytes l0a l0b l1a l1b l2a l2b under a 16-bit tag; 16-B load I
    (granules 2 I, 2 I + 1) becomes dword I of each plane, so that a plane's octet reads (0 4 1 5 | 2 6 3 7) — the order in which
    `v & 0x0F0F0F0F` / `(v >> 4) & 0x0F0F0F0F` leave a Q4 stream dword's nibbles (q4_levels_to_stream: nibble p <-> j = 2 (p & 3) + (p >> 2))."""
    K = x.shape[0]
    assert K % 8 == 0
    limbs = f8_limbs(x).reshape(3, K // 8, 8).astype(np.uint64)
    gran = np.zeros((K // 8, 4), dtype=np.uint64)
    for j in range(4):
        a, b = limbs[:, :, j], limbs[:, :, j + 4]
        gran[:, j] = (a[0] | (b[0] << 8) | (a[1] << 16) | (b[1] << 24) | (a[2] << 32) | (b[2] << 40) | (np.uint64(tag & 0xFFFF) << 48))
    gran = gran.reshape(-1)
    planes = np.zeros((3, K), dtype=np.uint8)
    lo = (gran & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (gran >> np.uint64(32)).astype(np.uint32)
    for i in range(K // 4):  # 16-B load i = granules 2 i, 2 i + 1 (v0 = lo[2i], v1 = hi[2i], v2 = lo[2i+1], v3 = hi[2i+1])
        v0, v1, v2, v3 = int(lo[2 * i]), int(hi[2 * i]), int(lo[2 * i + 1]), int(hi[2 * i + 1])
        words = ((v0 & 0xFFFF) | ((v2 & 0xFFFF) << 16), (v0 >> 16) | (v2 & 0xFFFF0000), (v1 & 0xFFFF) | ((v3 & 0xFFFF) << 16))
        for c in range(3):
            planes[c, 4 * i:4 * i + 4] = np.frombuffer(np.uint32(words[c]).tobytes(), dtype=np.uint8)
    return gran, planes
