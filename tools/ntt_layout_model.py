"""Index-level model of the 16-coefficients-per-lane NTT kernels (csrc/fz_ntt.hip).

Pure-Python emulation of the data movement and twiddle indexing of the two-pass kernels
(strided pass with wave-uniform twiddles / LDS transpose / contiguous pass with per-lane
twiddles).  Used to check the decomposition against the Longa-Naehrig loops before a
kernel goes to the GPU, and kept as executable documentation (tests/test_layout_model.py).
Arithmetic here is plain Python integers; the kernels do the same steps in exact fp64.
"""


def cent(x, q):
    y = x % q
    return y - q if y > q // 2 else y


def bitrev(i, k):
    r = 0
    for b in range(k):
        r |= ((i >> b) & 1) << (k - 1 - b)
    return r


def table(root, q, n):
    k = n.bit_length() - 1
    return [pow(root, bitrev(i, k), q) for i in range(n)]


def fwd_two_pass(x, q, root):
    """natural in -> bit-reversed out; mirrors ntt_fwd16<LOGD>."""
    D = len(x)
    logd = D.bit_length() - 1
    assert 5 <= logd <= 8
    L, SB = D // 16, logd - 4
    tw = table(root, q, D)
    # lane r (of L) holds a[k] = x[r + L*k]
    regs = [[x[r + L * k] for k in range(16)] for r in range(L)]
    for r in range(L):
        a = regs[r]
        for s in range(4):
            tk = 8 >> s
            for k in range(16):
                if k & tk:
                    continue
                w = tw[(1 << s) + (k >> (4 - s))]      # wave-uniform
                u, v = a[k], a[k + tk] * w
                a[k], a[k + tk] = (u + v) % q, (u - v) % q
    # transpose through "LDS": y[j], j = r + L*k
    y = [0] * D
    for r in range(L):
        for k in range(16):
            y[r + L * k] = regs[r][k]
    out = [0] * D
    for b in range(L):
        c = y[16 * b:16 * b + 16]
        for ls in range(SB):
            t = 1 << (SB - 1 - ls)
            ng = 16 // (2 * t)
            for k in range(16):
                if k & t:
                    continue
                g = k >> (SB - ls)
                w = tw[(16 << ls) + b * ng + g]          # per-lane table entry
                u, v = c[k], c[k + t] * w
                c[k], c[k + t] = (u + v) % q, (u - v) % q
        out[16 * b:16 * b + 16] = [cent(v, q) for v in c]
    return out


def inv_two_pass(x, q, inv_root):
    """bit-reversed in -> natural out incl. n^-1; mirrors ntt_inv16<LOGD>."""
    D = len(x)
    logd = D.bit_length() - 1
    assert 5 <= logd <= 8
    L, SB = D // 16, logd - 4
    itw = table(inv_root, q, D)
    n_inv = pow(D, q - 2, q)
    y = [0] * D
    for b in range(L):
        c = list(x[16 * b:16 * b + 16])
        for ls in range(SB):
            t = 1 << ls
            ng = 8 >> ls
            for k in range(16):
                if k & t:
                    continue
                g = k >> (ls + 1)
                w = itw[(D >> (ls + 1)) + b * ng + g]
                u, v = c[k], c[k + t]
                c[k], c[k + t] = (u + v) % q, ((u - v) * w) % q
        y[16 * b:16 * b + 16] = c
    out = [0] * D
    for r in range(L):
        a = [y[r + L * k] for k in range(16)]
        for s in range(4):
            tk = 1 << s
            h = 8 >> s
            for k in range(16):
                if k & tk:
                    continue
                w = itw[h + (k >> (s + 1))]
                u, v = a[k], a[k + tk]
                if s == 3:   # fold n^-1 into the last stage
                    a[k], a[k + tk] = ((u + v) * n_inv) % q, ((u - v) * (w * n_inv % q)) % q
                else:
                    a[k], a[k + tk] = (u + v) % q, ((u - v) * w) % q
        for k in range(16):
            out[r + L * k] = cent(a[k], q)
    return out


def fwd_small(x, q, root):
    """thread-per-polynomial kernel (D <= 16): plain LN loop."""
    a = list(x)
    n = len(a)
    tw = table(root, q, n)
    t, m = n, 1
    while m < n:
        t //= 2
        for i in range(m):
            for j in range(2 * i * t, 2 * i * t + t):
                u, v = a[j], a[j + t] * tw[m + i]
                a[j], a[j + t] = (u + v) % q, (u - v) % q
        m *= 2
    return [cent(v, q) for v in a]


def fwd_radix4(x, q, root):
    """4 coefficients per lane, log4(D) in-place passes; mirrors ntt_fwd4<LOGD>."""
    D = len(x)
    logd = D.bit_length() - 1
    assert logd in (6, 8)
    P, LP = logd // 2, D // 4
    tw = table(root, q, D)
    mem = list(x)                       # the LDS image (natural positions; the swizzle is a relabelling)
    for i in range(P):
        s = D >> (2 * i + 2)
        nxt = list(mem)
        for mm in range(LP):
            base = (mm // s) * 4 * s + mm % s
            a = [mem[base + k * s] for k in range(4)]
            g, pw = mm // s, 1 << (2 * i)
            wA, wB0, wB1 = tw[pw + g], tw[2 * pw + 2 * g], tw[2 * pw + 2 * g + 1]
            for (lo, hi, w) in ((0, 2, wA), (1, 3, wA), (0, 1, wB0), (2, 3, wB1)):
                u, v = a[lo], a[hi] * w
                a[lo], a[hi] = (u + v) % q, (u - v) % q
            for k in range(4):
                nxt[base + k * s] = a[k]
        mem = nxt
    return [cent(v, q) for v in mem]


def inv_radix4(x, q, inv_root):
    """mirrors ntt_inv4<LOGD> (n^-1 folded into the final stage)."""
    D = len(x)
    logd = D.bit_length() - 1
    assert logd in (6, 8)
    P, LP = logd // 2, D // 4
    itw = table(inv_root, q, D)
    n_inv = pow(D, q - 2, q)
    mem = list(x)
    for i in range(P):
        s = 1 << (2 * i)
        nxt = list(mem)
        for mm in range(LP):
            base = (mm // s) * 4 * s + mm % s
            a = [mem[base + k * s] for k in range(4)]
            g = mm // s
            wA0, wA1, wB = itw[D // (2 * s) + 2 * g], itw[D // (2 * s) + 2 * g + 1], itw[D // (4 * s) + g]
            last = i == P - 1
            for (lo, hi, w, fin) in ((0, 1, wA0, False), (2, 3, wA1, False), (0, 2, wB, last), (1, 3, wB, last)):
                u, v = a[lo], a[hi]
                if fin:
                    a[lo], a[hi] = ((u + v) * n_inv) % q, ((u - v) * (w * n_inv % q)) % q
                else:
                    a[lo], a[hi] = (u + v) % q, ((u - v) * w) % q
            for k in range(4):
                nxt[base + k * s] = a[k]
        mem = nxt
    return [cent(v, q) for v in mem]
