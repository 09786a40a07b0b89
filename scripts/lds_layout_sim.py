#!/usr/bin/env python3
"""LDS bank-conflict model of the fused FFT passes (csrc/ra_fft.cuh) for candidate slot layouts.

Rules from MI355X_MICROARCH.md (LDS): ds_read_b128 is served in four 16-lane groups, bank = 16-byte slot mod 16;
ds_write_b128 in eight groups of 8 contiguous lanes, slot mod 8.  A group costs as many cycles as the most loaded bank has
distinct addresses.  Prints, per layout, LDS cycles per wave-instruction relative to the conflict-free count.

    python scripts/lds_layout_sim.py
"""
import itertools

READ_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]
WRITE_GROUPS = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def brev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def cost(slots, groups, mod):
    """slots: per lane slot index or None (inactive)."""
    cyc = 0
    for g in groups:
        banks = {}
        for l in g:
            s = slots[l]
            if s is None:
                continue
            banks.setdefault(s % mod, set()).add(s)
        cyc += max((len(v) for v in banks.values()), default=0) if banks else 0
    return cyc


def passes(log2M):
    """(log2h, S) of every fused pass, like ra_fft_fused."""
    out, done = [], 0
    first = min(3, log2M)
    out.append((0, first))
    done = first
    while done < log2M:
        S = min(3, log2M - done)
        out.append((done, S))
        done += S
    return out


def simulate(log2M, T, slot):
    M = 1 << log2M
    tot_r = tot_w = ideal_r = ideal_w = 0
    for pi, (log2h, S) in enumerate(passes(log2M)):
        groups = M >> S
        h = 1 << log2h
        R = 1 << S
        for wave0 in range(0, min(groups, 4096), 64):
            lanes = list(range(wave0, wave0 + 64))
            for k in range(R):
                sl = []
                for gi in lanes:
                    if gi >= groups:
                        sl.append(None)
                        continue
                    g = brev(gi, log2M - S) if (pi == 0 and log2M > S) else gi
                    r = g & (h - 1)
                    base = ((g - r) << S) + r
                    sl.append(slot(base + k * h))
                n_act = sum(s is not None for s in sl)
                if pi > 0:
                    tot_r += cost(sl, READ_GROUPS, 16)
                    ideal_r += (n_act + 15) // 16
                tot_w += cost(sl, WRITE_GROUPS, 8)
                ideal_w += (n_act + 7) // 8
    # real-spectrum split: lanes read Z(k) and Z(M - k)
    for wave0 in range(0, M // 2 + 1, 64):
        ks = [k for k in range(wave0, wave0 + 64)]
        for f in (lambda k: k, lambda k: (M - k) % M):
            sl = [slot(f(k)) if k <= M // 2 else None for k in ks]
            tot_r += cost(sl, READ_GROUPS, 16)
            ideal_r += (sum(s is not None for s in sl) + 15) // 16
    return tot_r, ideal_r, tot_w, ideal_w


def layouts(log2M):
    M = 1 << log2M
    yield "identity", lambda e: e
    yield "pad e>>4 (current)", lambda e: e + (e >> 4)
    yield "pad e>>3", lambda e: e + (e >> 3)
    for a, b in itertools.product(range(3, 10), repeat=2):
        if a < b:
            yield f"xor3(e>>{a}) ^ xor3(e>>{b})", (lambda a, b: lambda e: e ^ ((e >> a) & 7) ^ ((e >> b) & 7))(a, b)
    for a in range(3, 10):
        yield f"xor3(e>>{a})", (lambda a: lambda e: e ^ ((e >> a) & 7))(a)
        yield f"xor4(e>>{a})", (lambda a: lambda e: e ^ ((e >> a) & 15))(a)
    top = max(log2M - 3, 3)
    yield "xor3(top3)", lambda e: e ^ ((e >> top) & 7)
    for a in range(3, 8):
        yield f"xor3(top3) ^ xor4(e>>{a})", (lambda a: lambda e: e ^ ((e >> top) & 7) ^ ((e >> a) & 15))(a)
        yield f"xor3(top3) ^ xor3(e>>{a})", (lambda a: lambda e: e ^ ((e >> top) & 7) ^ ((e >> a) & 7))(a)
        yield f"pad e>>4 + xor3(top3 of e)", lambda e: (e ^ ((e >> top) & 7)) + (e >> 4)


def main():
    for log2M, T in ((8, 64), (11, 256), (12, 512), (13, 512)):
        print(f"--- M = {1 << log2M}")
        res = []
        seen = set()
        for name, f in layouts(log2M):
            if name in seen:
                continue
            seen.add(name)
            r, ir, w, iw = simulate(log2M, T, f)
            res.append(((r + w) / (ir + iw), r / ir, w / iw, name))
        res.sort()
        for tot, r, w, name in res[:8]:
            print(f"  total x{tot:.2f}  reads x{r:.2f}  writes x{w:.2f}   {name}")
        for tot, r, w, name in res:
            if "current" in name or name == "identity":
                print(f"  total x{tot:.2f}  reads x{r:.2f}  writes x{w:.2f}   {name}")


if __name__ == "__main__":
    main()


def rule(log2M):
    """the layout csrc/ra_fft.cuh uses: ra_zslot"""
    zs = log2M - 3 if log2M >= 7 else 31
    return lambda e: e ^ ((e >> zs) & 7) ^ ((e >> 3) & 15)


def check_rule():
    print("--- ra_zslot over all sizes")
    for log2M in range(1, 14):
        M = 1 << log2M
        f = rule(log2M)
        img = sorted(f(e) for e in range(M))
        assert img == list(range(M)) or (M < 16 and len(set(img)) == M and max(img) < 16), (log2M, img[:20])
        r, ir, w, iw = simulate(log2M, 256, f)
        r0, _, w0, _ = simulate(log2M, 256, lambda e: e + (e >> 4))
        print(f"  M = {M:5d}: reads x{r / ir:.2f} writes x{w / iw:.2f} total x{(r + w) / (ir + iw):.2f}   (pad e>>4: total x{(r0 + w0) / (ir + iw):.2f})")


if __name__ == "__main__":
    check_rule()
