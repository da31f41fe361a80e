"""What bounds the time of one batch of 4096?  A model with two measured rates - a lone wave needs T1 = 41 us per
iteration, a SIMD with two or more waves issues TS = 27 us of iteration work per iteration (the bulk rate) - and the
iteration counts of the BASELINE batch from the CPU oracle (cap 60), run under the placements the GPU can or could
use.  Reproduces the measured batch times (profiles/r02_residency*.txt) and gives the bounds no placement beats.
CPU only:  python tools/sim_schedule.py > profiles/r02_schedule_sim.txt"""
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
T1, TS, NSIMD = 41.0, 27.0, 1024


def iteration_counts(cap=60):
    import mpc_rl_for_avs_amd  # noqa: F401
    from mpc_rl_for_avs_amd import synth
    from mpc_rl_for_avs_amd.reference_path import reference_states
    import oracle_lib
    inp = synth.solver_inputs(4096, 8, seed=0)
    r = oracle_lib.solve_batch(reference_states(0.1), inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                               vref=inp["vref"], others=inp["others"], collision_cost=True, max_iter=cap, xy_bounds=False)
    return r["iters"].astype(float)


def rate(n):
    return 1.0 / T1 if n == 1 else (1.0 / TS) / n


def dispatch(jobs, resident):
    """the hardware's placement: at most `resident` waves per SIMD, blocks dealt in launch order to free slots; the waves
    of a SIMD share its issue slots equally"""
    rem = [[] for _ in range(NSIMD)]
    q = list(jobs)[::-1]
    for _ in range(resident):
        for s in range(NSIMD):
            if q:
                rem[s].append(q.pop())
    t = 0.0
    active = set(s for s in range(NSIMD) if rem[s])
    while active:
        dt = min(min(rem[s]) / rate(len(rem[s])) for s in active)
        t += dt
        for s in list(active):
            d = dt * rate(len(rem[s]))
            keep = [x - d for x in rem[s] if x - d > 1e-9]
            for _ in range(len(rem[s]) - len(keep)):
                if q:
                    keep.append(q.pop())
            rem[s] = keep
            if not keep:
                active.discard(s)
    return t


def cu_pool(jobs, waves=8):
    """one workgroup per CU holding 16 instances in LDS, `waves` waves (2 per SIMD) that each run ONE iteration of the
    ready instance with the most iterations done, then hand it back; the first wave of a SIMD is favoured by the
    issue arbiter (runs at T1), its mate gets the rest"""
    tb = 1.0 / (1.0 / TS - 1.0 / T1)
    n = len(jobs)
    rem, done, ready = list(jobs), [0] * n, set(range(n))
    now, heap, busy_a, idle = 0.0, [], [False] * 4, []

    def start(w):
        if not ready:
            if w < 4:
                busy_a[w] = False
            idle.append(w)
            return
        j = max(ready, key=lambda i: done[i])
        ready.discard(j)
        if w < 4:
            busy_a[w] = True
            dur = T1
        else:
            dur = tb if busy_a[w % 4] else T1
        heapq.heappush(heap, (now + dur, w, j))

    for w in range(waves):
        start(w)
    while heap:
        now, w, j = heapq.heappop(heap)
        rem[j] -= 1
        done[j] += 1
        if rem[j] > 0:
            ready.add(j)
        start(w)
        while idle and ready:
            start(idle.pop())
    return now


def cu_migrate(jobs):
    """16 waves per CU as now (oldest wave of a SIMD runs at T1, the next gets the rest, the others wait), plus: whenever
    the SIMDs of the CU differ by two or more waves the youngest instance of the fullest moves to the emptiest"""
    r1, r2 = 1.0 / T1, 1.0 / TS - 1.0 / T1
    simd = [[] for _ in range(4)]
    for i, j in enumerate(jobs):
        simd[i % 4].append(float(j))
    now = 0.0
    while any(simd):
        dts = [s[0] / r1 for s in simd if s] + [s[1] / r2 for s in simd if len(s) >= 2]
        dt = min(dts)
        now += dt
        for s in simd:
            if s:
                s[0] -= dt * r1
            if len(s) >= 2:
                s[1] -= dt * r2
            s[:] = [x for x in s if x > 1e-9]
        while True:
            ln = [len(s) for s in simd]
            a, b = int(np.argmax(ln)), int(np.argmin(ln))
            if ln[a] - ln[b] < 2:
                break
            simd[b].append(simd[a].pop())
    return now


def main():
    it = iteration_counts()
    print(f"4096 instances, cap 60: iterations mean {it.mean():.2f}, {int((it >= 60).sum())} at the cap, "
          f"{int((it >= 35).sum())} with 35 or more")
    print(f"bounds: total work / {NSIMD} SIMDs = {it.sum() * TS / NSIMD:.0f} us, longest chain = {it.max() * T1:.0f} us")
    for resident, what in ((4, "all resident, 4 per SIMD (the kernel as it ships)"),
                           (3, "3 per SIMD resident, the rest dispatched as slots free"),
                           (2, "2 per SIMD resident")):
        print(f"{what}: as given {dispatch(it, resident):.0f} us, longest first {dispatch(np.sort(it)[::-1], resident):.0f}, "
              f"longest last {dispatch(np.sort(it), resident):.0f}")
    pools = [cu_pool(it[c * 16:(c + 1) * 16]) for c in range(256)]
    print(f"CU-wide pool, instances handed between 8 waves per iteration: {max(pools):.0f} us (mean CU {np.mean(pools):.0f})")
    mig = [cu_migrate(it[c * 16:(c + 1) * 16]) for c in range(256)]
    print(f"16 waves per CU with migration to idle SIMDs of the CU: {max(mig):.0f} us (mean CU {np.mean(mig):.0f})")


if __name__ == "__main__":
    main()
