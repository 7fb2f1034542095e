#!/usr/bin/env python
"""Discrete-event model of the persistent rollout scheduler (csrc/vf_persistent.h).

Workgroup slots draw items in ticket order; an item starts when its slot is free AND the
producer phases have finished the samples it covers.  Item durations are the measured per-item
run times (tools/persist_stats.py, B=200, 128-row tiles).  Used to compare ticket orders
(phase-major, sample cohorts with a phase offset, diagonal sample skew) before building them.

    python tools/sched_sim.py
"""
import heapq

import numpy as np

# name, items per step at B=200, samples per item, duration us, deps
STEP = [
    ('sa',     50, 4, 4.0,   ['LAST']),
    ('enc0', 1600, 1, 22.0,  ['LAST']),
    ('lstm1', 1600, 1, 224., ['enc0']),
    ('lstm2', 1600, 1, 214., ['lstm1']),
    ('enc1',  400, 1, 55.,   ['lstm2']),
    ('lstm3', 800, 1, 308.,  ['enc1']),
    ('lstm4', 800, 1, 412.,  ['lstm3']),
    ('enc2',  200, 2, 96.,   ['lstm4']),
    ('enc3',  200, 2, 21.,   ['enc2', 'sa']),
    ('lstm5', 400, 2, 510.,  ['enc3']),
    ('convt1', 400, 2, 95.,  ['lstm5']),
    ('lstm6', 800, 1, 570.,  ['convt1']),
    ('convt2', 800, 1, 76.,  ['lstm6']),
    ('lstm7', 1600, 1, 300., ['convt2']),
    ('fc',    256, 0, 143.,  ['lstm5']),      # 0: covers every sample of its cohort
    ('fin',   200, 1, 19.,   ['fc']),
    ('convt3', 1600, 1, 53., ['lstm7']),
    ('comp', 3200, 1, 22.7,  ['convt3', 'fin']),
]
B_REF = 200


def build(B, steps, cohorts=1, fc_groups=1):
    """-> list of cohorts; each cohort = list of phases dicts in program order."""
    out = []
    for c in range(cohorts):
        b0, b1 = B * c // cohorts, B * (c + 1) // cohorts
        nb = b1 - b0
        phases = []
        last = None
        for s in range(steps):
            ids = {}
            for name, items, spi, dur, deps in STEP:
                if spi == 0:
                    n_items, cover = max(items * nb // B_REF, 8), None
                else:
                    per_sample = items * spi / B_REF
                    n_units = (nb + spi - 1) // spi
                    n_items = int(round(per_sample * n_units))
                    cover = spi
                d = [last if x == 'LAST' else ids[x] for x in deps]
                ph = dict(name=name, step=s, cohort=c, n_items=n_items, cover=cover, dur=dur,
                          deps=[x for x in d if x is not None], b0=b0, nb=nb, idx=len(phases))
                ids[name] = len(phases)
                phases.append(ph)
            last = ids['comp']
        out.append(phases)
    return out


def simulate(cohort_phases, order, slots=512, overhead=3.0):
    """order: list of (cohort, phase_idx) in ticket order.  Returns (makespan_ms, wait_ms_per_slot)."""
    # completion time per (cohort, phase, sample) = max over its items
    done = {}
    free = [0.0] * slots
    heapq.heapify(free)
    total_wait = 0.0
    for c, pi in order:
        ph = cohort_phases[c][pi]
        nb = ph['nb']
        if ph['cover'] is None:
            per_item_samples = None
        n = ph['n_items']
        fin = np.zeros(nb)
        dep_ready = np.zeros(nb)
        for d in ph['deps']:
            dep_ready = np.maximum(dep_ready, done[(c, d)])
        if ph['cover'] is None:
            ready_all = dep_ready.max() if nb else 0.0
        items_per_unit = n / max((nb + (ph['cover'] or nb) - 1) // (ph['cover'] or nb), 1)
        for i in range(n):
            t = heapq.heappop(free)
            if ph['cover'] is None:
                ready = ready_all
                lo, hi = 0, nb
            else:
                unit = int(i // items_per_unit)
                lo = min(unit * ph['cover'], nb - 1)
                hi = min(lo + ph['cover'], nb)
                ready = dep_ready[lo:hi].max()
            start = max(t, ready)
            total_wait += start - t
            end = start + ph['dur'] + overhead
            fin[lo:hi] = np.maximum(fin[lo:hi], end)
            heapq.heappush(free, end)
        done[(c, pi)] = fin
    makespan = max(free)
    return makespan / 1e3, total_wait / slots / 1e3


def order_phase_major(cp):
    return [(0, i) for i in range(len(cp[0]))]


def order_cohort_offset(cp, offset):
    order, pos = [], [0] * len(cp)
    vt = 0
    while any(pos[c] < len(cp[c]) for c in range(len(cp))):
        for c in range(len(cp)):
            i = vt - c * offset
            if i >= 0 and i == pos[c] and pos[c] < len(cp[c]):
                order.append((c, i))
                pos[c] += 1
        vt += 1
    return order


def simulate_ready_queue(phases, slots=512, overhead=3.0):
    """Ideal list scheduling: a free slot takes the READY item with the smallest ticket (no slot is
    ever held by a waiting item).  One cohort, phases in program order.  Returns makespan in ms."""
    # build items: (ticket, phase idx, lo, hi)
    items = []
    for pi, ph in enumerate(phases):
        nb = ph['nb']
        n = ph['n_items']
        if ph['cover'] is None:
            for i in range(n):
                items.append((len(items), pi, 0, nb))
        else:
            units = (nb + ph['cover'] - 1) // ph['cover']
            per_unit = n / units
            for i in range(n):
                u = int(i // per_unit)
                lo = min(u * ph['cover'], nb - 1)
                items.append((len(items), pi, lo, min(lo + ph['cover'], nb)))
    # remaining producer items per (phase, sample)
    remaining = {}
    for (tk, pi, lo, hi) in items:
        for b in range(lo, hi):
            remaining[(pi, b)] = remaining.get((pi, b), 0) + 1
    # consumers waiting on (phase, sample)
    waiting_on = {}
    unmet = [0] * len(items)
    for (tk, pi, lo, hi) in items:
        for d in phases[pi]['deps']:
            for b in range(lo, hi):
                if remaining.get((d, b), 0) > 0:
                    waiting_on.setdefault((d, b), []).append(tk)
                    unmet[tk] += 1
    ready = [tk for tk in range(len(items)) if unmet[tk] == 0]
    heapq.heapify(ready)
    running = []            # (end_time, ticket)
    now, free = 0.0, slots
    done = 0
    while done < len(items):
        while free > 0 and ready:
            tk = heapq.heappop(ready)
            heapq.heappush(running, (now + phases[items[tk][1]]['dur'] + overhead, tk))
            free -= 1
        end, tk = heapq.heappop(running)
        now = end
        free += 1
        done += 1
        _, pi, lo, hi = items[tk]
        for b in range(lo, hi):
            remaining[(pi, b)] -= 1
            if remaining[(pi, b)] == 0:
                for c in waiting_on.pop((pi, b), []):
                    unmet[c] -= 1
                    if unmet[c] == 0:
                        heapq.heappush(ready, c)
    return now / 1e3


def main():
    steps = 13
    base = build(200, steps)
    ms, wait = simulate(base, order_phase_major(base))
    work = sum(p['n_items'] * (p['dur'] + 3.0) for p in base[0]) / 512 / 1e3
    print('work/slot %.1f ms' % work)
    print('phase-major, 1 cohort : makespan %.1f ms  wait/slot %.1f ms' % (ms, wait))
    print('ideal ready queue     : makespan %.1f ms  (no slot ever held by a waiting item)' % simulate_ready_queue(base[0]))
    for cohorts in (2, 3, 4, 8):
        for offset in (3, 6, 9, 12, 18, 27):
            cp = build(200, steps, cohorts)
            ms, wait = simulate(cp, order_cohort_offset(cp, offset))
            print('cohorts %d offset %2d     : makespan %.1f ms  wait/slot %.1f ms' % (cohorts, offset, ms, wait))


if __name__ == '__main__':
    main()
