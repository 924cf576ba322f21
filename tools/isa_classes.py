#!/usr/bin/env python3
"""tools/isa_classes.py -- where the vector instructions of the headline kernel go, by ISSUE-RATE CLASS and weighted by how often
each instruction runs per tile, from the compiler's own assembly of k_frame_p<5, true, 1, false> (METHOD, PACKED, VEC, SPREAD) --
the packed-once kernel of round 5 (mlvfs_amd/csrc/k_frame_p.hip + k_frame_dev.h); rounds 2-4 analysed k_frame<5, true, 1, false>.

How (no GPU needed):
 1. compiles k_frame_p.hip twice with the production flags -- once plainly, once with -gline-tables-only -- and checks that the
    kernel's instruction stream is the same in both (line tables must not change code generation);
 2. walks the annotated assembly: every instruction carries its source location INCLUDING the inlined-at chain
    (".loc ... ; k_frame_p.hip:L @[ k_frame_dev.h:L' @[ ... ] ]");
 3. gives every instruction a weight = wave-executions per tile on the benchmark's frames (common path), from WHERE in the source
    it sits -- found by searching the source for the statements that delimit each region, so the table survives edits (weight_of);
 4. classes every instruction by mnemonic and prices the class with the measured rates (profiles/r01/valu_rate2.log,
    profiles/r03/valu_rate3.log, profiles/r05/valu_rate4.log; unknown mnemonics at the quarter rate);
 5. checks the weighted total against SQ_INSTS_VALU of the same build when given.

usage: python tools/isa_classes.py [--pmc-valu-per-frame N] [--clock-ghz 2.35] [--json out.json]
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mlvfs_amd", "csrc")
SRC = os.path.join(CSRC, "k_frame_p.hip")
DEV = os.path.join(CSRC, "k_frame_dev.h")
KERNEL = "_ZN3mlv9k_frame_pILi5ELb1ELi1ELb0EEEvNS_9FrameArgsE"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"),
         "-mllvm", "--amdgpu-sched-strategy=max-ilp", "-fno-slp-vectorize", "-DKFP_ONLY"]
W, H, TCW, TCH5 = 3584, 1320, 64, 15
TILES_PER_FRAME = ((W // 2 + TCW - 1) // TCW) * ((H // 2 + TCH5 - 1) // TCH5)


def compile_asm(out_dir, debug):
    tag = "g" if debug else "prod"
    d = os.path.join(out_dir, tag)
    os.makedirs(d, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *(["-gline-tables-only"] if debug else []), "--save-temps=obj", "-c", SRC, "-o", os.path.join(d, "k_frame_p.o")]
    subprocess.run(cmd, check=True, capture_output=True, cwd=CSRC)
    text = open(os.path.join(d, "k_frame_p-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    start = text.index("\n" + KERNEL + ":")
    end = text.index(".Lfunc_end", start)
    return text[start:end].splitlines()


INSTR = re.compile(r"^\t([a-z][a-z0-9_]+)\b(.*)$")


def instruction_stream(lines):
    out = []
    for ln in lines:
        m = INSTR.match(ln)
        if m and not ln.startswith("\t."):
            ops = re.sub(r"\.L[A-Za-z0-9_]+", "L", m.group(2).split(";")[0]).strip()
            out.append((m.group(1), ops))
    return out


# ---------------------------------------------------------------- source regions
def _index(path):
    src = open(path).read().splitlines()

    def find(text, nth=1):
        for i, l in enumerate(src, 1):
            if text in l:
                nth -= 1
                if nth == 0:
                    return i
        raise SystemExit(f"isa_classes: marker not found in {os.path.basename(path)}: {text!r}")

    def block(text, nth=1):                     # from the marker's line to the brace that closes the first '{' at or after it
        a = find(text, nth)
        depth, seen = 0, False
        for i in range(a - 1, len(src)):
            for ch in src[i].split("//")[0]:
                if ch == "{":
                    depth += 1
                    seen = True
                elif ch == "}":
                    depth -= 1
                    if seen and depth == 0:
                        return a, i + 1
        raise SystemExit("isa_classes: unbalanced braces")
    return find, block


def build_regions():
    find, block = _index(SRC)
    dfind, dblock = _index(DEV)
    P = {"loop": block("    while (t < band_end) {"), "skip": block("        if (fb_skip > 0) {"),
         "sample": block("        if (!cont) {", 1), "top": block("        if (!cont) {", 2),
         "patch_fetch": block("        if (a.patch) {"), "patch": block("        if (tile_patched) {", 2),
         "draw": (find("else draw(nt, ne);", 2),) * 2, "loader": block("        if (tid_o < N_ITEMS) {"),
         "prefetch": (find("Pos nxt = pos_below(cur);"), find("const OutArgs oa = out_args(cold_args());")),
         "medians": (find("int tid_m = tid;"), find("if (is_strip && smooth_row && pk_uncertain(o))")),
         "output": block("        if (is_strip) {"), "carry": (find("int4 carry = make_int4(0, 0, 0, 0);"), find("par ^= 1;")),
         "uncertain": block("        if (unc) {")}
    D = {n: dblock(sig) for n, sig in {"dark": "void cell_multi_ev_dark(", "pair": "void cell_pair_ev(", "stripe_px": "uint32_t stripe_px(",
                                        "stripe_strip": "void stripe_strip(uint32_t", "fetch_clamped": "uint32_t fetch_clamped("}.items()}
    D["issue_general"] = (dfind("// (issue_item's arithmetic: rows and groups clamped to the frame)"), dfind("asm volatile(\"\" : \"+v\"(oa0), \"+v\"(ob0));"))
    # the output stage's instantiations, by the line of their call in strip_output: (CLAMP, XM) and the all-stripes one
    for name, args in {"out_generic": "true, true, true, false, SM>", "out_low_xm": "true, true, false, false, SM>", "out_low": "true, false, false, false, SM>",
                       "out_xm": "false, true, false, false, SM>", "out_plain": "false, false, false, false, SM>",
                       "out_bright": "false, false, false, true, SM>"}.items():
        D[name] = (dfind("strip_output_t<METHOD, PACKED, VECST, " + args),) * 2
    return P, D


def inside(line, rng):
    return rng[0] <= line <= rng[1]


# tiles of the benchmark's launch that do NOT continue the tile above them (first tiles of runs, tiles drawn singly, column tops)
FIRST_TILE_SHARE = 0.11
BORDER_TILE_SHARE = 1 - (26 / 28) * (43 / 44)          # tiles whose new rows or halo columns leave the frame (general prefetch form)
LOW_TILE_SHARE = 0.15                                   # tiles that hold a pixel at most 64 above black or pixel-map cells (with the tile above)
MARGIN_TILE_SHARE = 2 / 28                              # tiles at the frame's left or right margin


def weight_of(chain, R, tiles_per_workgroup):
    """chain: [(is_dev_header, line)] from the innermost frame to the outermost (the line in k_frame_p's own body is last)."""
    P, D = R
    if not chain:
        return 4.0, "unattributed"
    dev = [l for d, l in chain if d]
    own = [l for d, l in chain if not d]
    o = own[-1] if own else 0
    if any(inside(l, D[k]) for l in dev for k in ("dark", "pair")):
        return 0.0, "loader: pixels at or below black / beyond the table (not on these frames)"
    if any(inside(l, D[k]) for l in dev for k in ("stripe_px", "stripe_strip")):
        return 0.0, "stripes, 32-bit / generic epilogue (not this launch)"
    if not inside(o, P["loop"]):
        return 4.0 / tiles_per_workgroup, "prologue / epilogue (per workgroup)"
    if inside(o, P["skip"]) or inside(o, P["uncertain"]):
        return 0.0, "tiles listed for k_frame (none on these frames)"
    if inside(o, P["patch_fetch"]) or inside(o, P["patch"]):
        return 0.4, "pixel-map cells (one tile in ten has any; all four waves)"
    if inside(o, P["top"]):
        return 2.0 * FIRST_TILE_SHARE, "loader: the four rows above a run's first tile (waves 0-1, one tile in nine)"
    if inside(o, P["sample"]):
        return 4.0 * FIRST_TILE_SHARE, "reference sample of a run's first tile"
    if inside(o, P["draw"]):
        return 0.25 * FIRST_TILE_SHARE, "drawing the next run (one wave)"
    if inside(o, P["loader"]):
        return 4.0, "loader: items of 4 cells, packed against the tile's reference (all waves)"
    if inside(o, P["prefetch"]):
        if any(inside(l, D["issue_general"]) for l in dev):
            return 4.0 * BORDER_TILE_SHARE, "prefetch: tiles at the frame's border (general form)"
        if any(inside(l, D["fetch_clamped"]) for l in dev):
            return 4.0 * FIRST_TILE_SHARE, "reference sample of a run's first tile"
        return 4.0, "prefetch of the next tile, coordinates"
    if inside(o, P["medians"]):
        return 4.0, "medians: packed-once neighbour-sharing chain, hand-over, certainty test"
    if inside(o, P["output"]):
        if any(inside(l, D["out_generic"]) for l in dev):
            return 0.0, "stripes, 32-bit / generic epilogue (not this launch)"
        # (the benchmark's frames: every tile without low pixels or pixel-map cells that is not at the margin holds pixels >= 256 above black only)
        for name, share, what in (("out_bright", (1 - LOW_TILE_SHARE) * (1 - MARGIN_TILE_SHARE), "bright tiles: no clamp, no stripes mask, no conditions"),
                                  ("out_plain", 0.0, "no clamp, no stripes mask"),
                                  ("out_low", LOW_TILE_SHARE * (1 - MARGIN_TILE_SHARE), "tiles with low pixels or pixel-map cells"),
                                  ("out_xm", (1 - LOW_TILE_SHARE) * MARGIN_TILE_SHARE, "tiles at the frame's left / right margin"),
                                  ("out_low_xm", LOW_TILE_SHARE * MARGIN_TILE_SHARE, "margin tiles with low pixels")):
            if any(inside(l, D[name]) for l in dev):
                return 4.0 * share, "output stage: look-ups, R / B replacement, stripes, stores -- " + what
        return 4.0, "output stage: what the variants share (green EVs, EV sums, dispatch)"
    if inside(o, P["carry"]):
        return 4.0, "rows handed down to the tile below, end of tile"
    return 4.0, "loop control, tile bookkeeping"


# ---------------------------------------------------------------- instruction classes
def load_rates():
    rates = {}
    for rel in ("profiles/r01/valu_rate2.log", "profiles/r03/valu_rate3.log", "profiles/r05/valu_rate4.log"):
        p = os.path.join(ROOT, rel)
        if not os.path.exists(p):
            continue
        for ln in open(p):
            m4 = re.match(r"(v_[a-z0-9_]+)([^\n]*?)\s+([0-9.]+) sequences/clk/SIMD .* clk per sequence of 1$", ln.rstrip())
            if m4:                                              # (valu_rate4's format: single-instruction rows only)
                key = m4.group(1) + ("_dpp" if "row_" in m4.group(2) or "wave_" in m4.group(2) else "")
                if m4.group(1) != "v_cndmask_b32":
                    rates.setdefault(key, float(m4.group(3)))
                continue
            m = re.match(r"(v_[a-z0-9_]+)(.*?)thr=\s*(\d+)\s+([0-9.]+) wave-instr", ln)
            if m and int(m.group(3)) == 1024:
                key = m.group(1) + ("_dpp" if "row_" in m.group(2) or "wave_" in m.group(2) else "")
                if m.group(1) == "v_cndmask_b32":
                    if rel.endswith("valu_rate2.log"):
                        continue                               # superseded by valu_rate3 (see its header)
                    key = "v_cndmask_b32:vcc" if "vcc" in m.group(2) else "v_cndmask_b32:sgpr"
                rates.setdefault(key, float(m.group(4)))
    return rates


def class_of(mn, ops, rates):
    """-> (class name, rate in wave-instructions per clock and SIMD) for vector-ALU instructions; None for the rest"""
    if not mn.startswith("v_"):
        return None
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    if base == "v_cndmask_b32":                          # the VCC form and the SGPR-pair form were timed separately (valu_rate3)
        r = rates.get("v_cndmask_b32:vcc" if mn.endswith("_e32") or mn == base and "vcc" in ops else "v_cndmask_b32:sgpr")
        if r is not None:
            return ("slow (v_cndmask_b32 on VCC: measured 0.044)" if r < 0.15 else "quarter-rate (min/max/med3, packed, bfe, cvt, mul, cmp, 3-operand)"), r
    if mn.endswith("_dpp") or " row_" in ops or "wave_sh" in ops or "quad_perm" in ops:
        key = base + "_dpp"
        r = rates.get(key, rates.get("v_mov_b32_dpp", 0.245))
        return "quarter-rate (DPP operand)", min(r, 0.25)
    r = rates.get(base)
    if r is None:
        return "quarter-rate (not timed: assumed)", 0.245
    if r >= 0.40:
        return "half-rate (v_mov_b32, f32 add/sub/mul, unpacked 16-bit min/max)", r
    if r >= 0.28:
        return "third-rate (add/sub, logic, constant shifts)", r
    if r < 0.15:
        return "slow (measured below 0.15)", r
    return "quarter-rate (min/max/med3, packed, bfe, cvt, mul, cmp, 3-operand)", r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc-valu-per-frame", type=float, default=None, help="SQ_INSTS_VALU per frame of the same build (cross-check)")
    ap.add_argument("--clock-ghz", type=float, default=2.35, help="in-kernel clock measured on the box (DESIGN.md 3.1)")
    ap.add_argument("--frames-per-launch", type=int, default=100)
    ap.add_argument("--workgroups", type=int, default=1024)
    ap.add_argument("--json", default=None)
    ap.add_argument("--out-dir", default="/tmp/mlvfs_amd_isa")      # (not under the repository: the temporaries are 100 MB and would travel with gpurun)
    a = ap.parse_args()

    prod = instruction_stream(compile_asm(a.out_dir, False))
    dbg_lines = compile_asm(a.out_dir, True)
    dbg = instruction_stream(dbg_lines)
    same = prod == dbg
    R = build_regions()
    rates = load_rates()
    tiles_per_wg = TILES_PER_FRAME * a.frames_per_launch / a.workgroups

    loc_re = re.compile(r"(k_frame_p\.hip|k_frame_dev\.h):(\d+):\d+")
    chain = []
    per_region, per_class, unknown = {}, {}, {}
    counts = {"valu": 0.0, "salu": 0.0, "lds": 0.0, "vmem": 0.0, "static_valu": 0, "static_all": 0}
    for ln in dbg_lines:
        if ln.startswith("\t.loc"):
            found = [(f.endswith(".h"), int(x)) for f, x in loc_re.findall(ln.split(";", 1)[1] if ";" in ln else "") if int(x) > 0]
            if found:                                          # (a location without a line in k_frame.hip is compiler-made, "line 0":
                chain = found                                  #  such instructions stay with the code around them)
            continue
        m = INSTR.match(ln)
        if not m or ln.startswith("\t."):
            continue
        mn, ops = m.group(1), m.group(2)
        w, region = weight_of(chain, R, tiles_per_wg)
        counts["static_all"] += 1
        c = class_of(mn, ops, rates)
        if c is None:
            if mn.startswith("ds_"):
                counts["lds"] += w
            elif mn.startswith(("buffer_", "global_", "flat_", "scratch_")):
                counts["vmem"] += w
            elif mn.startswith("s_"):
                counts["salu"] += w
            continue
        counts["static_valu"] += 1
        counts["valu"] += w
        name, rate = c
        reg = per_region.setdefault(region, {"weight": w, "valu": 0.0, "clk": 0.0, "static": 0})
        reg["valu"] += w
        reg["clk"] += w / rate
        reg["static"] += 1
        cl = per_class.setdefault(name, {"valu": 0.0, "clk": 0.0, "mnemonics": {}})
        cl["valu"] += w
        cl["clk"] += w / rate
        cl["mnemonics"][mn] = cl["mnemonics"].get(mn, 0.0) + w
        if "not timed" in name:
            unknown[mn] = unknown.get(mn, 0.0) + w

    simds = 1024
    tiles = TILES_PER_FRAME
    flat_clk = counts["valu"] / 0.24
    weighted_clk = sum(c["clk"] for c in per_class.values())
    us = lambda clk_per_tile: clk_per_tile * tiles / simds / (a.clock_ghz * 1e3)
    res = {
        "kernel": "k_frame_p<5, true, 1, false>", "same_instruction_stream_with_line_tables": same,
        "static_instructions": counts["static_all"], "static_valu": counts["static_valu"],
        "tiles_per_frame": tiles, "valu_wave_instructions_per_tile": round(counts["valu"], 1),
        "valu_wave_instructions_per_frame": round(counts["valu"] * tiles),
        "lds_instructions_per_tile": round(counts["lds"], 1), "vmem_instructions_per_tile": round(counts["vmem"], 1),
        "salu_instructions_per_tile": round(counts["salu"], 1),
        "clock_ghz": a.clock_ghz,
        "flat_floor_us_per_frame_at_0.24": round(us(flat_clk), 2),
        "class_weighted_floor_us_per_frame": round(us(weighted_clk), 2),
        "classes": {k: {"valu_per_tile": round(v["valu"], 1), "share": round(v["valu"] / counts["valu"], 3),
                        "clk_per_tile": round(v["clk"]), "top": dict(sorted(((m, round(x, 1)) for m, x in v["mnemonics"].items()), key=lambda t: -t[1])[:8])}
                    for k, v in sorted(per_class.items(), key=lambda t: -t[1]["valu"])},
        "regions": {k: {"wave_executions_per_tile": v["weight"] if v["weight"] >= 0.1 else round(v["weight"], 4), "static_valu": v["static"],
                        "valu_per_tile": round(v["valu"], 1), "share": round(v["valu"] / counts["valu"], 3)}
                    for k, v in sorted(per_region.items(), key=lambda t: -t[1]["valu"])},
        "not_timed_mnemonics": dict(sorted(((m, round(x, 1)) for m, x in unknown.items()), key=lambda t: -t[1])),
    }
    if a.pmc_valu_per_frame:
        res["pmc_SQ_INSTS_VALU_per_frame"] = a.pmc_valu_per_frame
        res["static_model_over_pmc"] = round(counts["valu"] * tiles / a.pmc_valu_per_frame, 3)
    text = json.dumps(res, indent=1)
    if a.json:
        open(a.json, "w").write(text + "\n")
    print(text)
    if not same:
        print("isa_classes: WARNING: the instruction stream differs between the production build and the -gline-tables-only build", file=sys.stderr)


if __name__ == "__main__":
    main()
