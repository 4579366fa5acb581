"""CPU emulation of select_groups_kernel's radix select (csrc/topk.hip kth_largest) on the group maxima of one query:
how many 8-bit passes it takes, how many bins the first pass populates and how large the largest bin is (= LDS atomics on one
address), without and with the lower bound (the smallest per-thread maximum).  Scores are drawn as N(0, 1/512) — what a random
unit query gives over N = 118 287 unit rows; `--negative-group` makes one group of 16 rows all-negative, the case that put the
first digit at the sign bit ("The 6 us that looked like hardware and were data")."""
import sys
import numpy as np

def okey(f):
    u = f.view(np.uint32).astype(np.uint64)
    return np.where((u & 0x80000000) != 0, (~u) & 0xffffffff, u | 0x80000000).astype(np.uint64)

def passes(lst, k):
    aor, aand = np.bitwise_or.reduce(lst), np.bitwise_and.reduce(lst)
    hb = int(aor ^ aand).bit_length() - 1
    mask = 0 if hb == 63 else (~((2 << hb) - 1)) & (2**64 - 1)
    prefix, need, sh = int(aand) & mask, k, max(hb - 7, 0)
    width, out, done = hb - sh + 1, [], False
    while not done:
        dm = (1 << width) - 1
        sel = lst[(lst & np.uint64(mask)) == np.uint64(prefix)]
        h = np.bincount(((sel >> np.uint64(sh)) & np.uint64(dm)).astype(np.int64), minlength=256)
        run = 0
        for b in range(255, -1, -1):
            if run < need and need <= run + h[b]:
                done = h[b] == need - run; need -= run; prefix |= b << sh
                break
            run += h[b]
        out.append(f"digit at bit {sh + width - 1}: {len(sel)} keys in {int((h > 0).sum())} bins, largest {int(h.max())}")
        mask |= dm << sh
        if sh == 0:
            break
        nsh = max(sh - 8, 0); width = sh - nsh; sh = nsh
    return out

N, k = 118287, 100
rng = np.random.default_rng(0)
for q in range(4):
    sc = (rng.standard_normal(N) / np.sqrt(512)).astype(np.float32)
    if "--negative-group" in sys.argv:
        g0 = 16 * int(rng.integers(0, N // 16))
        sc[g0:g0 + 16] = -np.abs(sc[g0:g0 + 16])
    keys = (okey(sc) << np.uint64(32)) | ((~np.arange(N, dtype=np.uint64)) & np.uint64(0xffffffff))
    pad = np.zeros((N + 15) // 16 * 16, np.uint64); pad[:N] = keys
    gm = pad.reshape(-1, 16).max(1)
    g8 = np.zeros(8192, np.uint64); g8[:len(gm)] = gm                    # thread t holds keys t, t + 1024, ...
    tmax = g8.reshape(8, 1024).max(0)
    lb = tmax[tmax > 0].min()
    print(f"query {q}: all keys: " + " | ".join(passes(gm, k)))
    print(f"          keys >= the smallest thread maximum ({int((gm >= lb).sum())} of {len(gm)}): " + " | ".join(passes(gm[gm >= lb], k)))
