#!/usr/bin/env python3
"""Ablation timing of k_frame<5,...> via MLVFS_AMD_DBG bits (1 no planes, 2 no medians, 4 no stores, 8 no patches)."""
import os, sys, subprocess
for dbg in (0, 8, 1, 2, 3, 4, 7, 15):
    env = dict(os.environ, MLVFS_AMD_DBG=str(dbg), KB_ROUNDS="4")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "kbench.py")], env=env, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("m5  ")]
    print(f"dbg={dbg:2d}", line[0] if line else out[-300:])
