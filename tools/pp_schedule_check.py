"""Interval model of the wave-role-split 256 x 256 loop (kronfluence_amd/csrc/kf_pingpong.h): checks the RAW / WAR ordering of
the LDS-DMA requests against the fragment reads for every request schedule (`ISSUE` 0 / 1 / 2) and loop length.

The model is the header's ordering argument made executable.  Eight waves meet at raw s_barriers; group X (waves 0-3) runs its
k-th segment in the interval S_k between barrier events E_k and E_k+1, group Y (waves 4-7, one extra barrier up front) its
(k-1)-th.  Segments of one k-tile t: L(2t) M(2t) L(2t+1) M(2t+1).  Within an interval anything may interleave, so

  * a request issued by a group in interval i and covered by that group's counted `s_waitcnt vmcnt(N)` at the END of its segment
    in interval w may write LDS at any time in [i, w]  (vmcnt retires in order: the wait covers all but the last N requests);
  * a fragment read in interval r has returned by the end of r (every L segment ends with lgkmcnt(0) before its barrier).

A piece (A0 / A1 / B0 / B1 of a k-tile) is staged by BOTH groups (each wave moves its share).  Reading piece p of k-tile t in
interval r is correct iff every group's request for (p, t) has w < r (RAW) and every group's request for (p, t + 2) -- the next
occupant of the same LDS rows -- has i > r (WAR), and the request for (p, t) was issued after every read of (p, t - 2).

This transcribes the schedule of the header by hand (it cannot read the C++); `tests/test_host_logic.py` runs it, the race
screens of `tests/test_ops_gpu.py` run the kernels themselves.
"""
import sys

PIECES = ("A0", "A1", "B0", "B1")


def program(issue, nt):
    """The segments of ONE group in program order: list of (kind, reads, issues, vmcnt) with reads / issues = [(piece, tile)],
    vmcnt = the counted wait at the end of the segment (None: no wait).  Segment 0 is the prologue."""
    segs = []
    pro = [("A0", 0), ("B0", 0), ("B1", 0), ("A1", 0)]
    if issue == 0:
        if nt > 1:
            segs.append(("P", [], pro + [("A0", 1), ("B0", 1)], 6))
        else:
            segs.append(("P", [], pro, 2))
    else:
        if nt > 1:
            segs.append(("P", [], pro + [("A0", 1), ("B0", 1), ("B1", 1)], 8))
        else:
            segs.append(("P", [], pro, 2))
    for t in range(nt):
        more1, more2 = t + 1 < nt, t + 2 < nt
        r0 = [("A0", t), ("B0", t), ("B1", t)]
        r1 = [("A1", t)]
        if issue == 0:
            segs.append(("L", r0, [("B1", t + 1), ("A1", t + 1)] if more1 else [], 8 if more1 else 0))
            segs.append(("M", [], [], None))
            segs.append(("L", r1, [("A0", t + 2), ("B0", t + 2)] if more2 else [], 6 if more2 else (2 if more1 else None)))
            segs.append(("M", [], [], None))
        else:
            segs.append(("L", r0, [], 6 if more1 else 0))
            segs.append(("M", [], [("A1", t + 1)] if more1 else [], None))
            early = [("A0", t + 2)] if (issue == 2 and more2) else []
            wait = None
            if more1:
                wait = 4 if (issue == 2 and more2) else 2
            segs.append(("L", r1, early, wait))
            late = []
            if more2:
                late = ([] if issue == 2 else [("A0", t + 2)]) + [("B0", t + 2), ("B1", t + 2)]
            segs.append(("M", [], late, None))
    return segs


def windows(segs, offset):
    """Per request (piece, tile) of one group: (issue interval, interval of the wait that covers it).  `offset` = interval of
    the group's segment 1 minus 1 (X: 0, Y: 1); the prologue of both groups is interval -1 (before E_0)."""
    out, queue = {}, []   # queue: requests in flight, oldest first; each request = 2 DMA instructions
    for k, (_, _, issues, vmcnt) in enumerate(segs):
        interval = -1 if k == 0 else k - 1 + offset
        for req in issues:
            assert req not in out, f"{req} issued twice"
            out[req] = [interval, None]
            queue.append(req)
        if vmcnt is not None:
            assert vmcnt % 2 == 0
            keep = vmcnt // 2
            done, queue = (queue[:-keep], queue[-keep:]) if keep else (queue, [])
            for req in done:
                out[req][1] = interval
    assert not queue, f"requests never waited for: {queue}"
    return out


def reads(segs, offset):
    out = {}
    for k, (_, rd, _, _) in enumerate(segs):
        for req in rd:
            out.setdefault(req, []).append(k - 1 + offset)
    return out


def check(issue, nt):
    segs = program(issue, nt)
    win = {g: windows(segs, off) for g, off in (("X", 0), ("Y", 1))}
    rds = {g: reads(segs, off) for g, off in (("X", 0), ("Y", 1))}
    errors = []
    for t in range(nt):
        for p in PIECES:
            for g in ("X", "Y"):
                if (p, t) not in win[g]:
                    errors.append(f"{p}({t}) never staged by {g}")
    for g in ("X", "Y"):
        for (p, t), rs in rds[g].items():
            for r in rs:
                for h in ("X", "Y"):
                    i, w = win[h][(p, t)]
                    if not w < r:
                        errors.append(f"RAW: {g} reads {p}({t}) in S_{r}, {h}'s request (issued S_{i}) is only waited for in S_{w}")
                    if (p, t + 2) in win[h] and not win[h][(p, t + 2)][0] > r:
                        errors.append(f"WAR: {g} reads {p}({t}) in S_{r}, {h} issues {p}({t + 2}) in S_{win[h][(p, t + 2)][0]}")
    # latency allowance: segments between the issue and the covering wait (a wait in the issuing segment exposes the whole latency)
    slack = min(w - i for g in win for (p, t), (i, w) in win[g].items() if t > 0)if nt > 1 else None
    return errors, slack


def main():
    bad = 0
    for issue in (0, 1, 2):
        slacks = []
        for nt in range(1, 9):
            errors, slack = check(issue, nt)
            slacks.append(slack)
            for e in errors:
                bad += 1
                print(f"ISSUE {issue} nt {nt}: {e}")
        print(f"ISSUE {issue}: k-tile counts 1..8 checked, min intervals between issue and covering wait {[s for s in slacks if s is not None]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
