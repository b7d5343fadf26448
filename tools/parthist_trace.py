#!/usr/bin/env python3
"""Summary of a SYBL_PARTHIST_TRACE file (csrc/kernels.hip: k_part_hist's phase timestamps, 100 MHz wall clock):
per-phase durations, how long a workgroup's waves wait for the slowest one, gaps between the workgroups of a compute unit.
usage: parthist_trace.py <file>"""
import sys
import numpy as np
t = np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.int64)
t0 = t[:, 0].min()
us = lambda x: x / 100.0
ph = {"regions": t[:, 1] - t[:, 0], "walk": t[:, 2] - t[:, 1], "epilogue": t[:, 4] - t[:, 2], "whole": t[:, 4] - t[:, 0]}
print("items", len(t), " kernel span %.1f us" % us(t[:, 4].max() - t0))
for k, v in ph.items():
    print("%-14s mean %8.1f  min %8.1f  max %8.1f us" % (k, us(v.mean()), us(v.min()), us(v.max())))
wv = t[:, 16:32]
wv = np.where(wv == 0, t[:, 2:3], wv)
print("wave walk end: first-to-last spread mean %.1f us, max %.1f us; mean idle per wave %.1f us" % (
    us((wv.max(1) - wv.min(1)).mean()), us((wv.max(1) - wv.min(1)).max()), us((wv.max(1)[:, None] - wv).mean())))
# per compute unit: (xcc, se, cu) from HW_ID
hw = t[:, 5]
cu = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 0x7) * 64 + ((hw >> 8) & 0xF)
gaps, busy = [], []
for c in np.unique(cu):
    rows = t[cu == c]
    rows = rows[np.argsort(rows[:, 0])]
    busy.append((rows[:, 4] - rows[:, 0]).sum())
    gaps += list(rows[1:, 0] - rows[:-1, 4])
print("compute units seen", len(np.unique(cu)), " workgroups per CU min/max", np.bincount(np.unique(cu, return_inverse=True)[1]).min(), np.bincount(np.unique(cu, return_inverse=True)[1]).max())
print("gap between consecutive workgroups of a CU: mean %.1f us, max %.1f us" % (us(np.mean(gaps)) if gaps else 0, us(np.max(gaps)) if gaps else 0))
print("first start spread %.1f us; last end - first end %.1f us" % (us(np.sort(t[:, 0])[min(255, len(t) - 1)] - t0), us(t[:, 4].max() - t[:, 4].min())))
