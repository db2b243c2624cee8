"""orb_bf_query_packed_noalloc (the brute-force scan that allocates nothing: its
per-chunk partial results live in a fixed block of the library's device image)
against orb_bf_query_packed_device, the entry point it is a variant of, through
the C ABI.  Integer results: bit-exact.  Two calls in a row on the same device,
the second with fewer queries and a shorter database, so that stale words of the
shared block from the first call would show; then the oracle on a slice."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _descriptors(torch, n, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)


def test_no_alloc_entry_point_equals_allocating_one(oracle):
    import torch

    from orb_slam2_amd import binding

    lib = binding.lib()
    bf = binding.BruteForce(0)
    for call, (nq, nd, base) in enumerate([(3000, 300_000, 0), (517, 70_001, 1000)]):
        db = _descriptors(torch, nd, 10 + call)
        q = _descriptors(torch, nq, 20 + call)
        # planted near-duplicates and exact ties so that best / second are not all "random" distances
        q[:64] = db[torch.arange(64, device="cuda") * 997 % nd]
        db[nd - 1] = db[5]
        q[64] = db[5]
        bf.set_database_device(db, base)
        want = torch.zeros(nq, dtype=torch.int64, device="cuda")
        got = torch.full((nq,), -1, dtype=torch.int64, device="cuda")
        bf.query_packed_device(q, want)
        bf.query_packed_noalloc(q, got)
        torch.cuda.synchronize()
        assert torch.equal(got, want), "call %d: %d of %d packed results differ" % (call, int((got != want).sum()), nq)
        # and both equal the CPU oracle on a slice
        sl = slice(0, 96)
        idx, d1, d2 = oracle.bf_top2(q[sl].cpu().numpy(), db.cpu().numpy())
        p = got[sl].cpu().numpy().view(np.uint64)
        np.testing.assert_array_equal((p & np.uint64(0xFFFFFFFF)).astype(np.int64), idx.astype(np.int64) + base)
        np.testing.assert_array_equal((p >> np.uint64(48)).astype(np.uint16), d1)
        np.testing.assert_array_equal(((p >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.uint16), d2)
    # empty shard: every query reports "none"
    bf.set_database_device(db[:0], 0)
    got.fill_(0)
    bf.query_packed_noalloc(q, got)
    torch.cuda.synchronize()
    assert (got.cpu().numpy().view(np.uint64) == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    # beyond the stated limit: refused before anything is launched
    nbig = binding.ORB_BF_NOALLOC_MAX_QUERIES + 1
    rc = lib.orb_bf_query_packed_noalloc(bf.h, binding._dp(q), nbig, binding._dp(got), None)
    assert rc == -4 and b"limit" in lib.orb_last_error()
    bf.close()
