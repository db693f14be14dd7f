"""Model of the LDS-DMA issue order of the deep-pipelined full-row K loop (pv_gemm.hip, pv_gemm_fullrow_kernel<NT, DPH>; the kernel carries the
same model as the constexpr function pv_fr_counts): prints, per K-tile and phase, how many pieces are issued after the ones the NEXT phase
reads - the immediates of the counted s_waitcnt vmcnt.  NP phases per K-tile, PW pieces per wave and W group (= NT / NP), two pieces of A.
python scripts/fullrow_vmcnt_model.py"""


def sim(NP, PW, nk, AP=2):      # AP: A pieces per wave and K-tile (2; 3 for the 160-row tile of round 6)
    seq = []

    def issue(lbl, cnt):
        seq.extend([lbl] * cnt)

    def younger(labels):
        return len(seq) - 1 - max(i for i, l in enumerate(seq) if l in labels)

    issue(("A", 0), AP)
    for g in range(NP):
        issue(("W", g, 0), PW)
    if nk >= 2:
        issue(("A", 1), AP)
        for g in range(NP - 1):
            issue(("W", g, 1), PW)
    waits = {("pro",): younger([("A", 0), ("W", 0, 0)])}
    for s in range(nk):
        for j in range(NP):
            if j == 0:
                if s + 1 < nk:
                    issue(("W", NP - 1, s + 1), PW)
            elif s + 2 < nk:
                if j == 1:
                    issue(("A", s + 2), AP)
                    issue(("W", 0, s + 2), PW)
                else:
                    issue(("W", j - 1, s + 2), PW)
            if j < NP - 1:
                waits[(s, j)] = younger([("W", j + 1, s)])
            elif s + 1 < nk:
                waits[(s, j)] = younger([("A", s + 1), ("W", 0, s + 1)])
    return waits


if __name__ == "__main__":
    for NT, NP in ((4, 2), (6, 2), (8, 2), (6, 3), (8, 4)):
        w = sim(NP, NT // NP, 8)
        print(f"N = {64 * NT}, {NP} phases of {8 * NT // NP} MFMAs, 8 K-tiles:  prologue {w[('pro',)]}  steady " +
              " ".join(str(w[(3, j)]) for j in range(NP)) + "  K-tile nk-2 " + " ".join(str(w[(6, j)]) for j in range(NP)) +
              "  K-tile nk-1 " + " ".join(str(w[(7, j)]) for j in range(NP - 1)))
    for NT in (4, 6):          # round 6: the 160-row tile stages three A pieces per wave
        w = sim(2, NT // 2, 8, AP=3)
        print(f"N = {64 * NT}, 160-row tile (3 A pieces), 2 phases:  prologue {w[('pro',)]}  steady " + " ".join(str(w[(3, j)]) for j in range(2)) +
              "  K-tile nk-2 " + " ".join(str(w[(6, j)]) for j in range(2)) + "  K-tile nk-1 " + str(w[(7, 0)]))
