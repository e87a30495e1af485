"""CPU restatement of the device random-number path (oracle; TEST INFRASTRUCTURE ONLY).

The reference draws its minibatch indices with `np.random.randint(0, size, B)` (utils/buffer.py:40) and its
Gaussian noise with torch's default generator (networks/vae.py:55-56, agent/sac/actor.py:47-60 through
`Normal.rsample`, agent/diffsrsac/diffsrsac_agent.py:276-283).  Neither generator exists on the device; the HIP
path uses the counter-based Philox4x32-10 of Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"
(SC'11) -- the Random123 library's `philox4x32_R(10, ctr, key)` -- so that a hipGraph replay draws fresh numbers
from a device-resident step counter.  This file restates (a) the Philox4x32-10 bijection, pinned by Random123's
published known-answer vectors (`KAT`, from Random123 `examples/kat_vectors`), and (b) the mapping from
(seed, offset, element index) to indices / normals that `rlrep_amd/csrc/elementwise.hip::philox_fill_body` and
`philox_index` implement, so that the GPU tests can check the device streams element by element.
"""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57            # Philox4x32 multipliers
W0, W1 = 0x9E3779B9, 0xBB67AE85            # Weyl key increments (golden ratio, sqrt(3) - 1)

# Random123 kat_vectors, rows "philox4x32 10 <ctr x4> <key x2> <expected x4>"
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(ctr, key):
    """ctr: uint32[..., 4], key: uint32[..., 2] (broadcastable) -> uint32[..., 4].  Ten rounds of
    (hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0) with the key bumped by the Weyl constants after every round."""
    c = np.array(ctr, dtype=np.uint64) & 0xFFFFFFFF
    k = np.array(key, dtype=np.uint64) & 0xFFFFFFFF
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = np.broadcast_to(k[..., 0], c0.shape).copy(), np.broadcast_to(k[..., 1], c0.shape).copy()
    for _ in range(10):
        p0 = M0 * c0                       # 32 x 32 -> 64 bit products fit a uint64
        p1 = M1 * c2
        hi0, lo0 = p0 >> 32, p0 & 0xFFFFFFFF
        hi1, lo1 = p1 >> 32, p1 & 0xFFFFFFFF
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def raw_stream(n, seed, offset, stream_id=0):
    """uint32[n]: element e = word e & 3 of block q = e >> 2, counter (q_lo, q_hi, off_lo, off_hi ^ stream_id),
    key (seed_lo, seed_hi)  -- the layout of philox_fill_body / philox_index."""
    nq = (n + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    off = np.uint64(offset & 0xFFFFFFFFFFFFFFFF)
    ctr = np.stack([q & 0xFFFFFFFF, q >> 32, np.full(nq, off & np.uint64(0xFFFFFFFF)),
                    np.full(nq, (off >> np.uint64(32)) ^ np.uint64(stream_id))], axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    return philox4x32_10(ctr, key).reshape(-1)[:n]


def indices(n, hi, seed, offset, stream_id=0):
    """Uniform integers in [0, hi): (word * hi) >> 32  (multiply-shift, no modulo bias beyond 2^-32 * hi)."""
    w = raw_stream(n, seed, offset, stream_id).astype(np.uint64)
    return ((w * np.uint64(hi)) >> np.uint64(32)).astype(np.int32)


def normals(n, std, seed, offset, stream_id=0):
    """float32[n]: Box-Muller on word pairs (0,1) and (2,3) of every block, u = ((w >> 8) + 0.5) / 2^24 in (0, 1);
    elements (2h, 2h+1) = r cos(2 pi u2), r sin(2 pi u2), r = sqrt(-2 ln u1), times std; all in float32."""
    nq = (n + 3) // 4
    w = raw_stream(4 * nq, seed, offset, stream_id).reshape(nq, 4)
    f32 = np.float32
    u = ((w >> np.uint32(8)).astype(f32) + f32(0.5)) * f32(1.0 / 16777216.0)
    out = np.empty((nq, 4), dtype=f32)
    for h in range(2):
        u1, u2 = u[:, 2 * h], u[:, 2 * h + 1]
        rad = np.sqrt(f32(-2.0) * np.log(u1)).astype(f32)
        ang = (f32(6.283185307179586) * u2).astype(f32)
        out[:, 2 * h] = rad * np.cos(ang).astype(f32) * f32(std)
        out[:, 2 * h + 1] = rad * np.sin(ang).astype(f32) * f32(std)
    return out.reshape(-1)[:n]
