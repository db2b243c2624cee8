"""Seeded synthetic inputs for parity tests and benchmarks.

No dataset ships with the reference or exists on the GPU box, so frames are
textured, corner-rich synthetic scenes with flat regions (so that the
minThFAST fallback of src/ORBextractor.cc:852-858 is exercised) and
descriptors are random 256-bit strings with planted near-duplicates.
All randomness comes from a counter-based SplitMix64, so any element can be
regenerated independently and results do not depend on numpy's generators.
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, n, offset=0):
    """n outputs of SplitMix64 started at `seed`, skipping `offset` outputs."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def frame_seed(config_id, frame_index):
    return 0x0B5EED00 + config_id * 1000003 + frame_index


def synth_scene(w, h, seed, shift=(0, 0), noise_seed=None):
    """u8 image h x w: mid-gray, floor(w*h/600) rectangles (side 6..48, gray
    16..240), floor(w*h/2400) discs (radius 3..20), painter's order, plus noise
    (sum of four uniform[0,7]) - 14 per pixel.  `shift` moves the shapes by
    (dx, dy) pixels, `noise_seed` (default: seed) seeds the noise alone."""
    nrect, ndisc = (w * h) // 600, (w * h) // 2400
    r = splitmix64(seed, nrect * 5 + ndisc * 4)
    img = np.full((h, w), 128, dtype=np.int16)
    dx, dy = shift
    rr = r[: nrect * 5].reshape(nrect, 5)
    rx = (rr[:, 0] % np.uint64(w)).astype(np.int64) + dx
    ry = (rr[:, 1] % np.uint64(h)).astype(np.int64) + dy
    rw = (rr[:, 2] % np.uint64(43)).astype(np.int64) + 6
    rh = (rr[:, 3] % np.uint64(43)).astype(np.int64) + 6
    rg = (rr[:, 4] % np.uint64(225)).astype(np.int64) + 16
    for i in range(nrect):
        x0, y0 = max(rx[i], 0), max(ry[i], 0)
        x1, y1 = min(rx[i] + rw[i], w), min(ry[i] + rh[i], h)
        if x1 > x0 and y1 > y0:
            img[y0:y1, x0:x1] = rg[i]
    dd = r[nrect * 5 :].reshape(ndisc, 4)
    cx = (dd[:, 0] % np.uint64(w)).astype(np.int64) + dx
    cy = (dd[:, 1] % np.uint64(h)).astype(np.int64) + dy
    cr = (dd[:, 2] % np.uint64(18)).astype(np.int64) + 3
    cg = (dd[:, 3] % np.uint64(225)).astype(np.int64) + 16
    for i in range(ndisc):
        x0, y0 = max(cx[i] - cr[i], 0), max(cy[i] - cr[i], 0)
        x1, y1 = min(cx[i] + cr[i] + 1, w), min(cy[i] + cr[i] + 1, h)
        if x1 <= x0 or y1 <= y0:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        m = (xx - cx[i]) ** 2 + (yy - cy[i]) ** 2 <= cr[i] ** 2
        img[y0:y1, x0:x1][m] = cg[i]
    nz = splitmix64((seed if noise_seed is None else noise_seed) ^ 0x5EEDFACE, w * h)
    s7 = np.uint64(7)
    noise = ((nz & s7) + ((nz >> np.uint64(3)) & s7) + ((nz >> np.uint64(6)) & s7) + ((nz >> np.uint64(9)) & s7)).astype(np.int16) - 14
    img += noise.reshape(h, w)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_descriptors(n, seed):
    """n random 256-bit descriptors as an (n, 32) u8 array."""
    return splitmix64(seed, n * 4).view(np.uint8).reshape(n, 32).copy()


def synth_queries(db, nq, seed, dup_frac=0.1, max_flips=60):
    """nq query descriptors: random, with a `dup_frac` share copied from rows
    of `db` and perturbed by 0..max_flips random bit flips."""
    q = synth_descriptors(nq, seed)
    ndup = int(nq * dup_frac)
    if ndup and len(db):
        r = splitmix64(seed ^ 0xD00D, ndup * 2 + ndup * max_flips)
        rows = (r[:ndup] % np.uint64(len(db))).astype(np.int64)
        flips = (r[ndup : 2 * ndup] % np.uint64(max_flips + 1)).astype(np.int64)
        bits = (r[2 * ndup :] % np.uint64(256)).astype(np.int64).reshape(ndup, max_flips)
        q[:ndup] = db[rows]
        for i in range(ndup):
            for b in bits[i, : flips[i]]:
                q[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return q


def synth_vocabulary(k, L, seed, prune=0.0, stop=0.0):
    """A vocabulary tree in the shape DBoW2 stores (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1368-1470):
    arrays over nodes 1..n in file order (breadth first, a parent before its children): parent id, leaf flag,
    32-byte descriptor, weight.  A child's descriptor is its parent's with random bits flipped, fewer the
    deeper the level, so the descent by Hamming distance is meaningful.  `prune` = fraction of inner nodes
    cut to a leaf early or given fewer than k children; `stop` = fraction of words with weight 0."""
    r = np.random.default_rng(seed)
    parent, leaf, desc, weight = [], [], [], []
    level_nodes = [(0, np.zeros(32, np.uint8))]  # (id, descriptor) of the current level; the root has none
    next_id = 1
    for depth in range(1, L + 1):
        new_level = []
        flips = max(4, 128 >> (depth - 1))
        for pid, pdesc in level_nodes:
            nchild = k
            if pid != 0 and prune > 0 and r.random() < prune:
                nchild = int(r.integers(1, k)) if k > 1 else 1
            for _ in range(nchild):
                bits = np.unpackbits(pdesc if pid else r.integers(0, 256, 32, dtype=np.uint8))
                idx = r.choice(256, size=flips, replace=False)
                bits[idx] ^= 1
                d = np.packbits(bits)
                is_leaf = depth == L or (prune > 0 and r.random() < prune * 0.5)
                parent.append(pid)
                leaf.append(1 if is_leaf else 0)
                desc.append(d)
                weight.append((0.0 if r.random() < stop else float(r.random() * 5 + 0.01)) if is_leaf else 0.0)
                if not is_leaf:
                    new_level.append((next_id, d))
                next_id += 1
        level_nodes = new_level
    return (np.array(parent, np.int32), np.array(leaf, np.uint8), np.array(desc, np.uint8).reshape(-1, 32),
            np.array(weight, np.float64))


def write_vocabulary_text(path, k, L, scoring, weighting, parent, leaf, desc, weight):
    """The text format of saveToTextFile (TemplatedVocabulary.h:1474-1497), the one ORBvoc.txt is in."""
    with open(path, "w") as f:
        f.write("%d %d  %d %d\n" % (k, L, scoring, weighting))
        for i in range(len(parent)):
            f.write("%d %d %s %s\n" % (parent[i], leaf[i], " ".join(str(int(b)) for b in desc[i]) + " ", repr(float(weight[i]))))
