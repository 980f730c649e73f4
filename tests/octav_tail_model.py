"""CPU model (numpy) of the EXACT-TAIL / BOUNDED-BULK walk of csrc/octav_tail.hpp — test infrastructure, never shipped
or called by the product.  It states, in a form a property test can hammer, the rule by which the kernel ACCEPTS a result
that was not obtained by evaluating every iterate of forward_net.py:323-330 exactly:

  histogram   64 bins per octave over 2^-18 .. 2^14, per bin an exact count and an exact sum (the kernel: integer mantissa
              sums) -> suffix totals N_ge[b], S_ge[b] = everything in bins >= b;
  list        every |x| whose bin is >= J (the gather threshold theta = lower edge of bin J, however it was chosen);
  bulk phase  while the iterate t lies in a bin b < J (and beyond, while more values lie above t than a wave's registers hold):
              t <- a LOWER BOUND of F(t) that needs only the suffix totals:
                  F(t) = (S_ge[b+1] + sum of the m values of bin b above t) / (c (n - N_ge[b+1] - m) + N_ge[b+1] + m),
              each of the m values lies in (t, edge[b+1]) and 0 <= m <= count[b]: the quotient is monotone in m once the
              values are put at t, so F(t) >= min(q(0), q(count[b])).  F is non-decreasing below its least fixed point
              (dropping a value v raises F iff v < (1 - c) F), hence by induction every bulk iterate stays BELOW the
              reference's iterate of the same index, and below the fixed point the reference ends on;
  exact phase once t is in a bin >= J: the reference's own step (oracle semantics: fp32 divide, |s' - s| < 1e-6 keeps s)
              with count / sum of the values above t taken exactly (suffix totals of the bins above + the listed values
              of t's bin);
  accept      only if every bulk step moved up by at least 2e-6 (the reference cannot have stopped there), no exact step
              moved down, the stop rule fired on an exact evaluation, there were >= 2 exact evaluations (or no bounded
              step at all) and <= 20 evaluations in all (the reference, which is never behind, then converged too).
Anything else is REJECTED: the kernel hands such a pair to the exact two-read rescue (bracket + gather + walk).
"""
import numpy as np

F32 = np.float32
LOG_NB = 2048
LOG_SHIFT = 17
LOG_KEY0 = (127 - 18) << 6
OCTAV_CONST = 1.0 / (4 ** 8) / 3.0


def log_bin_of_bits(bits):
    b = (np.asarray(bits, np.int64) >> LOG_SHIFT) - LOG_KEY0
    return np.clip(b, 0, LOG_NB - 1)


def log_bin(v):
    return int(log_bin_of_bits(np.asarray(F32(v)).view(np.uint32)))


def log_edge(b):
    if b <= 0:
        return F32(0.0)
    return np.asarray(np.uint32((b + LOG_KEY0) << LOG_SHIFT)).view(np.float32)[()]


class Hist:
    """What one read of the pair leaves behind."""

    def __init__(self, x):
        x = np.asarray(x, F32).ravel()
        self.n = x.size
        a = np.abs(x)
        bits = a.view(np.uint32).astype(np.int64)
        self.nan = bool(np.isnan(x).any())
        self.mn = F32(x.min()) if x.size else F32(np.inf)
        self.mx = F32(x.max()) if x.size else F32(-np.inf)
        key = bits >> LOG_SHIFT
        t = key - (LOG_KEY0 + 1)
        inw = (t >= 0) & (t < LOG_NB - 1)
        b = (t + 1)[inw]
        self.cnt = np.bincount(b, minlength=LOG_NB).astype(np.int64)
        self.sum = np.bincount(b, weights=a[inw].astype(np.float64), minlength=LOG_NB)
        out = ~inw & (bits != 0) & ~np.isnan(a)
        self.out_cnt = int(out.sum())
        self.out_sum = float(a[out].astype(np.float64).sum())
        self.high = bool((a[~np.isnan(a)] >= 16384.0).any())
        self.n_ge = np.concatenate([np.cumsum(self.cnt[::-1])[::-1], [0]])
        self.s_ge = np.concatenate([np.cumsum(self.sum[::-1])[::-1], [0.0]])
        self.abs = a
        self.bins = np.zeros(self.n, np.int64)
        self.bins[inw] = b

    def theta_bin(self, tau):
        """Largest bin J with at least tau * n values in the bins >= J (1 <= J)."""
        want = max(1, int(np.ceil(tau * self.n)))
        ok = np.nonzero(self.n_ge[:LOG_NB] >= want)[0]
        return int(ok.max()) if ok.size else 1


BULK_SHAVE = F32(0.99999952316284180)   # 1 - 2^-21: below the three fp32 roundings of a bound (csrc/octav_tail.hpp)


SURV_CAP = 1280   # values one wave's registers hold (kSurvCap): bounded steps go on until what lies above the iterate fits


def tail_walk(h, J, dynamic_sym=False, max_iters=20, surv_cap=SURV_CAP):
    """-> dict(status, s, evals, exact_evals, bulk_evals).  status: 'ok' (accepted), 'nan', or a rejection reason."""
    r = dict(status="ok", s=F32(np.nan), evals=0, exact_evals=0, bulk_evals=0)
    if h.n == 0:
        r["status"] = "empty"
        return r
    if h.nan:
        r["status"] = "nan"      # NaN is a fixed point of the reference's loop: finished, not rejected
        return r
    ud = 4.0 if (dynamic_sym and abs(float(h.mn)) < 1e-6) else 1.0
    c = OCTAV_CONST / ud
    nz = h.out_cnt + int(h.n_ge[1])
    with np.errstate(all="ignore"):
        s = F32(F32(h.out_sum + h.s_ge[1]) / F32(nz))
    if np.isnan(s):
        r["status"] = "nan"
        return r
    if h.high:
        r["status"] = "reject:above_window"
        return r
    n = h.n
    J = max(1, min(int(J), LOG_NB - 1))
    t = s
    evals = 0
    # ---- bulk phase: lower bounds from the suffix totals alone; they go on past the list's first bin while more values lie above
    # the iterate than one wave holds (the kernel then compacts them once and every exact step is cheap)
    fits = int(h.n_ge[J]) <= surv_cap
    while True:
        b = log_bin(t)
        if b < 1 or b > LOG_NB - 2:
            r["status"] = "reject:outside_window"
            return r
        A, nb1, nb = float(h.s_ge[b + 1]), int(h.n_ge[b + 1]), int(h.n_ge[b])
        if b >= J and (fits or nb <= surv_cap):
            break
        moved = False
        if nb1 != 0:
            q0 = F32(F32(A) / F32(c * (n - nb1) + nb1))
            qm = F32(F32(A + (nb - nb1) * float(t)) / F32(c * (n - nb) + nb))
            lb = F32(min(q0, qm) * BULK_SHAVE)
            moved = bool(F32(lb - t) >= F32(2e-6))
        if not moved:
            if b >= J:
                break                      # the bound has stopped moving inside the list: exact steps take over
            r["status"] = "reject:bulk_stall"
            return r
        t = lb
        evals += 1
        r["bulk_evals"] += 1
        if evals >= max_iters:
            r["status"] = "reject:cap_in_bulk"
            return r
    # ---- exact phase: the reference's step on exact totals
    s = t
    listed = h.abs[h.bins >= J]
    lbins = h.bins[h.bins >= J]
    while True:
        b = log_bin(s)
        if b < J:
            r["status"] = "reject:left_list"
            return r
        inb = listed[(lbins == b)]
        gt = inb[inb > s]
        cnt_gt = int(h.n_ge[b + 1]) + int(gt.size)
        tot = float(h.s_ge[b + 1]) + float(gt.astype(np.float64).sum())
        denom = c * (n - cnt_gt) + cnt_gt
        with np.errstate(all="ignore"):
            s1 = F32(F32(tot) / F32(denom))
        evals += 1
        r["exact_evals"] += 1
        if np.abs(F32(s1 - s)) < F32(1e-6):
            break
        if not (s1 >= s):
            r["status"] = "reject:decreased"
            return r
        s = s1
        if evals >= max_iters:
            r["status"] = "reject:cap"
            return r
    r["evals"] = evals
    if r["exact_evals"] < 2 and r["bulk_evals"] > 0:   # (a walk without a bounded step IS the reference's walk)
        r["status"] = "reject:one_exact"
        return r
    r["s"] = F32(s)
    return r
