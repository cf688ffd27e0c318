#!/usr/bin/env python3
"""Build-container only: turn the reference's recorded shark tracks (data/sharkTrackingData.csv: per shark four
rows x, vx, y, vy of 815 samples; SharkTrajectory takes them in that order, sharkTrajectory.py:7) into the data
fixture tests/golden/shark_tracking_xy.npz = positions only, [32 sharks, 815 samples, 2].  Data, not source."""
import csv
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("AUVP_REFERENCE", "/root/reference")
rows = np.array([[float(v) for v in r] for r in csv.reader(open(os.path.join(REF, "data", "sharkTrackingData.csv")))])
assert rows.shape == (128, 815), rows.shape
xy = np.stack([rows[0::4], rows[2::4]], axis=-1)  # [32, 815, 2]
# cross-check with the per-axis files robotSim.py:832 loads (same tracks in another frame: only the shape is checked)
xs = np.array([[float(v) for v in r] for r in csv.reader(open(os.path.join(REF, "data", "shark_tracking_data_x.csv")))])
assert xs.shape == (32, 815)
out = os.path.join(HERE, "shark_tracking_xy.npz")
np.savez_compressed(out, xy=xy, source="data/sharkTrackingData.csv rows 4k (x) and 4k+2 (y), k = shark id")
print("wrote", out, os.path.getsize(out), "bytes", xy.shape, xy[0, :3])
