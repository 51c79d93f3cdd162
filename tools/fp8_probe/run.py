import ctypes, torch
lib = ctypes.CDLL(__file__.rsplit("/", 1)[0] + "/libprobe.so")
p = lambda t: ctypes.c_void_p(t.data_ptr())
# A[m][k] = 1 for all; B[n][k] = 1 only in K-block kb (so the output isolates that block): C[m][n] = 32 * sa(m,kb) * sb(n,kb)
one = torch.ones(16, 128, device="cuda").to(torch.float8_e4m3fn)
for which in ("first", "second"):
    print("== scale of the", which, "operand")
    for kb in range(4):
        bsel = torch.zeros(16, 128, device="cuda"); bsel[:, 32 * kb:32 * kb + 32] = 1
        bsel = bsel.to(torch.float8_e4m3fn)
        hits = {}
        for L in range(64):
            for byte in range(4):
                x = torch.full((64,), 127 | 127 << 8 | 127 << 16 | 127 << 24, dtype=torch.int32, device="cuda")
                v = [127, 127, 127, 127]; v[byte] = 129
                w = v[0] | v[1] << 8 | v[2] << 16 | v[3] << 24; x[L] = w - (1 << 32) if w >= (1 << 31) else w
                u = torch.full((64,), 127 | 127 << 8 | 127 << 16 | 127 << 24, dtype=torch.int32, device="cuda")
                c = torch.zeros(16, 16, device="cuda")
                if which == "first": lib.probe_mfma(p(one), p(bsel), p(x), p(u), p(c))      # data rows of operand 1 <-> output rows m
                else: lib.probe_mfma(p(bsel), p(one), p(u), p(x), p(c))                       # K-block selected via operand 1, scales on operand 2 <-> output cols n
                d = (c != 32.0)
                if d.any():
                    rows = sorted(set(d.nonzero()[:, 0].tolist())); cols = sorted(set(d.nonzero()[:, 1].tolist()))
                    hits[(L, byte)] = (rows if len(rows) < 16 else "all", cols if len(cols) < 16 else "all", float(c[d].max()))
        ks = sorted(hits)
        print(" K-block", kb, ":", [(k, hits[k][0] if which == "first" else hits[k][1]) for k in ks][:20], "n_hits", len(ks))
