"""Reads the two charts the reference ships (step_response.png and control.png, drawn by src/main.rs:73-188) and
writes the plotted output curves as data under tests/golden/.

The PNGs are the only recorded outputs of the reference for its own scenario (dc_motor, dt = 0.05: a 10 s step
response, and 280 closed-loop MPC steps with f = v = 20).  The axes are calibrated from each chart's own major grid
lines, not from any number this repository computes, so the traces are an independent, if coarse (about +-1 pixel),
pin of the rollout and of the whole Phi / Gamma / H / gain chain behind the closed loop.

    python tools/extract_png_trace.py /root/reference
"""
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def extract(path, dy, dx, legend_box=None):
    im = np.array(Image.open(path).convert("RGB")).astype(int)
    grid = (np.abs(im[:, :, 0] - 184) < 3) & (np.abs(im[:, :, 1] - 184) < 3) & (np.abs(im[:, :, 2] - 184) < 3)
    axis = im.sum(axis=2) < 30
    x_axis_row = int(np.argmax(axis.sum(axis=1)))
    y_axis_col = int(np.argmax(axis.sum(axis=0)))
    # major grid lines are the rows / columns that are grid-coloured at (nearly) every probe position
    probes_c = [y_axis_col + d for d in (230, 251, 310, 330)]
    rows = [r for r in range(0, x_axis_row) if sum(bool(grid[r, c]) for c in probes_c) >= 3]
    probes_r = [rows[0] + d for d in (20, 27, 33)]
    cols = [c for c in range(y_axis_col + 1, im.shape[1]) if sum(bool(grid[r, c]) for r in probes_r) >= 2]
    rows_desc = sorted(rows, reverse=True)                 # bottom line is the lowest labelled value
    return im, x_axis_row, y_axis_col, rows_desc, cols


def trace(path, y_of_lowest_line, dy, dx, kmax, legend_box=None):
    im, x_axis_row, y_axis_col, rows_desc, cols = extract(path, dy, dx)
    yv = y_of_lowest_line + dy * np.arange(len(rows_desc))
    ky, cy = np.polyfit(yv, rows_desc, 1)                  # row = cy + ky * y
    xv = dx * np.arange(len(cols))
    kx, cx = np.polyfit(xv, cols, 1)                       # col = cx + kx * x
    green = (im[:, :, 1] > 120) & (im[:, :, 0] < 130) & (im[:, :, 2] < 140) & (im[:, :, 1] - im[:, :, 0] > 60)
    if legend_box:
        x0, x1, y0, y1 = legend_box
        green[y0:y1, x0:x1] = False
    top = min(rows_desc) - 80
    xs, lo, hi = [], [], []
    for c in range(y_axis_col + 1, im.shape[1]):
        x = (c - cx) / kx
        if x < 0 or x > kmax:
            continue
        rr = np.nonzero(green[max(top, 0):x_axis_row, c])[0] + max(top, 0)
        if rr.size == 0:
            continue
        xs.append(x)
        hi.append((rr.min() - 1.0 - cy) / ky)              # one pixel of slack for rounding / anti-aliasing
        lo.append((rr.max() + 1.0 - cy) / ky)
    return {"axis_fit": {"row_per_unit_y": ky, "row_at_y0": cy, "col_per_unit_x": kx, "col_at_x0": cx,
                         "y_grid_rows": rows_desc, "x_grid_cols": cols},
            "x": xs, "x_halfpixel": 0.5 / kx, "y_lo": lo, "y_hi": hi}


def main(ref_root):
    out = trace(os.path.join(ref_root, "step_response.png"), 0.0, 0.01, 20.0, 199)
    out["source"] = "step_response.png of the reference (src/main.rs:73-122): simulator::step output, 200 samples"
    dst = os.path.join(ROOT, "tests", "golden", "step_response_png_trace.json")
    json.dump(out, open(dst, "w"))
    print(len(out["x"]), "columns ->", dst)
    out = trace(os.path.join(ref_root, "control.png"), 0.0, 0.1, 20.0, 200, legend_box=(640, 795, 260, 335))
    out["source"] = ("control.png of the reference (src/main.rs:124-188): closed-loop `outputs` series (green); only the "
                     "stretches not overdrawn by the desired-trajectory and input series are visible")
    dst = os.path.join(ROOT, "tests", "golden", "control_png_trace.json")
    json.dump(out, open(dst, "w"))
    print(len(out["x"]), "columns ->", dst)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
