#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) as text.

    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    print('# rocprofv3 --kernel-trace --stats summary of %s' % path)
    print('# per kernel symbol: calls, total ms, average us, share of GPU time')
    for name, calls, total, avg, pct in cur.execute('select * from top_kernels'):
        print('%-110s %6d %10.3f ms %10.3f us %6.2f %%' % (name[:110], calls, total / 1e3, avg, pct))
    print('\n# per launch shape (grid in threads, LDS bytes): calls, average us')
    q = ('select name, grid_x, grid_y, grid_z, lds_size, count(*), avg(duration)/1000.0 from kernels '
         'group by name, grid_x, grid_y, grid_z, lds_size order by name, grid_x')
    for name, gx, gy, gz, lds, n, avg in cur.execute(q):
        print('%-60s grid %7d x %d x %2d  lds %6d  %5d calls  %10.3f us' % (name[:60], gx, gy, gz, lds, n, avg))


if __name__ == '__main__':
    main(sys.argv[1])
