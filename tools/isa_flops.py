#!/usr/bin/env python3
"""ISA-level accounting of one kernel instantiation of lib/libgigalens_hip.so (gfx950 code object).

What it does (no GPU needed; the same binaries run on the GPU box, so bench.py calls it live):
  1. pulls the gfx950 code object out of the shared library (.hip_fatbin -> clang-offload-bundler),
  2. reads the kernel's metadata (VGPRs, spills, scratch) from the AMDGPU notes,
  3. disassembles the kernel, splits it into basic blocks, finds the loops (backward branches) and their nesting,
  4. counts per block the VALU wave-instructions and the fp32 flops they perform per lane:
       v_pk_fma_f32 4 | v_pk_mul_f32 / v_pk_add_f32 2 | v_fma / v_fmac / v_mad 2 | v_mul / v_add / v_sub 1 |
       v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos 1 (quarter-rate transcendental, counted as one flop) |
       moves, selects, compares, min/max, conversions, integer and bit ops 0,
  5. turns the static counts into DYNAMIC counts per pixel with an execution model of the kernel's loop nest:
       flops per tile = (tile-loop body outside its inner loops) + sum_inner (body x trips) + (conditional tails x probability)
     The trip counts come from the caller (mean EPL series trip count of the batch, see `pair_model`).

The model is validated against hardware: the dynamic VALU wave-instruction count it predicts per launch is compared with the
PMC counter SQ_INSTS_VALU of the same launch (profiles/r2_isa_flops.md).

    python3 tools/isa_flops.py --kernel 'gl_pair_kernel<3, float __vector(2), 3, glk::KindList<1, 4>, glk::KindList<>, glk::KindList<16> >'
    python3 tools/isa_flops.py --list            # every kernel: VGPRs, spills, scratch
"""
import argparse
import atexit
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
LIB = os.path.join(ROOT, "gigalens_amd", "lib", "libgigalens_hip.so")

FLOPS = [  # (regex on the mnemonic, flops per lane)
    (r"^v_pk_fma_f32", 4), (r"^v_pk_(mul|add)_f32", 2),
    (r"^v_(fma|fmac|mad|mac)_f32", 2), (r"^v_fma_mix", 2),
    (r"^v_(mul|add|sub|subrev)_f32", 1), (r"^v_mul_legacy_f32", 1),
    (r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", 1), (r"^v_rcp_iflag_f32", 1), (r"^v_ldexp_f32", 1),
    (r"^v_(fma|mul|add)_f64", 0),  # fp64 is not on this path's hot kernels; kept out of the fp32 tally
]
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32|^v_rcp_iflag_f32")
PACKED = re.compile(r"^v_pk_(fma|mul|add)_f32")


def code_object(lib=LIB, workdir=None):
    """Paths of the gfx950 code objects extracted from `lib`: the library is several translation units, each contributing one
    offload bundle to the .hip_fatbin section."""
    if workdir is None:
        workdir = tempfile.mkdtemp(prefix="gl_isa_")
        atexit.register(shutil.rmtree, workdir, True)  # the extracted code objects are scratch: gone when the caller exits
    fat = os.path.join(workdir, "fatbin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(magic, blob)] + [len(blob)]
    cos = []
    for i in range(len(starts) - 1):
        part, co = os.path.join(workdir, f"bundle{i}"), os.path.join(workdir, f"gfx950_{i}.co")
        open(part, "wb").write(blob[starts[i]:starts[i + 1]])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        cos.append(co)
    return cos


def demangle(names):
    out = subprocess.run(["c++filt"] + list(names), capture_output=True, text=True, check=True).stdout
    return out.strip().split("\n")


def kernel_metadata(cos):
    """{demangled name: dict(symbol, co, vgpr_count, vgpr_spill_count, sgpr_spill_count, scratch_bytes, lds_bytes)}"""
    if isinstance(cos, str):
        cos = [cos]
    pat = re.compile(r"\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?"
                     r"\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?"
                     r"\.vgpr_spill_count:\s+(\d+)", re.S)
    out = {}
    for co in cos:
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True,
                             check=True).stdout
        rows = pat.findall(txt)
        if not rows:
            continue
        for d, r in zip(demangle([r[1] for r in rows]), rows):
            out[d] = dict(symbol=r[1], co=co, lds_bytes=int(r[0]), scratch_bytes=int(r[2]), sgpr_count=int(r[3]),
                          sgpr_spill_count=int(r[4]), vgpr_count=int(r[5]), vgpr_spill_count=int(r[6]))
    return out


def disassemble(co, symbol):
    """[(addr, mnemonic, operands)] of one kernel."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f"--disassemble-symbols={symbol}", co],
                         capture_output=True, text=True, check=True).stdout
    ins = []
    for line in txt.split("\n"):
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if mnem.startswith("s_cbranch") or mnem == "s_branch":
            t = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", line)
            base = ins[0][0] if ins else addr
            tgt = (base + int(t.group(1), 16)) if t else None
        ins.append((addr, mnem, ops, tgt))
    return ins


def weight(mnem):
    if not mnem.startswith("v_"):
        return 0
    for pat, f in FLOPS:
        if re.match(pat, mnem):
            return f
    return 0


class Loop:
    def __init__(self, header, blocks):
        self.header, self.blocks, self.children, self.parent = header, set(blocks), [], None

    @property
    def lo(self):
        return self.header


class CFG:
    """Basic blocks (split at branch targets and after branches), edges, dominators, natural loops."""

    def __init__(self, ins):
        self.ins = ins
        addrs = [i[0] for i in ins]
        leaders = {addrs[0]}
        for k, (addr, mnem, ops, tgt) in enumerate(ins):
            is_br = mnem.startswith("s_cbranch") or mnem == "s_branch"
            if is_br or mnem in ("s_endpgm", "s_setpc_b64"):
                if tgt is not None:
                    leaders.add(tgt)
                if k + 1 < len(ins):
                    leaders.add(addrs[k + 1])
        leaders = sorted(a for a in leaders if a in set(addrs))
        self.block_of = {}
        self.blocks = {}  # leader -> [instruction indices]
        cur = None
        lead = set(leaders)
        for k, a in enumerate(addrs):
            if a in lead:
                cur = a
                self.blocks[cur] = []
            self.blocks[cur].append(k)
            self.block_of[a] = cur
        self.succ = {b: [] for b in self.blocks}
        for b, idx in self.blocks.items():
            addr, mnem, ops, tgt = ins[idx[-1]]
            nxt = addrs[idx[-1] + 1] if idx[-1] + 1 < len(ins) else None
            if mnem == "s_branch":
                if tgt in self.block_of:
                    self.succ[b].append(self.block_of[tgt])
            elif mnem.startswith("s_cbranch"):
                if tgt in self.block_of:
                    self.succ[b].append(self.block_of[tgt])
                if nxt is not None:
                    self.succ[b].append(nxt)
            elif mnem in ("s_endpgm", "s_setpc_b64"):
                pass
            elif nxt is not None:
                self.succ[b].append(nxt)
        self.pred = {b: [] for b in self.blocks}
        for b, ss in self.succ.items():
            for s in ss:
                self.pred[s].append(b)
        self._dominators(addrs[0])
        self._loops()

    def _dominators(self, entry):
        nodes = list(self.blocks)
        dom = {n: set(nodes) for n in nodes}
        dom[entry] = {entry}
        changed = True
        while changed:
            changed = False
            for n in nodes:
                if n == entry:
                    continue
                ps = [dom[p] for p in self.pred[n]]
                new = (set.intersection(*ps) if ps else set()) | {n}
                if new != dom[n]:
                    dom[n], changed = new, True
        self.dom = dom

    def _loops(self):
        by_header = {}
        for u, ss in self.succ.items():
            for h in ss:
                if h in self.dom[u]:  # back edge u -> h
                    body, stack = {h, u}, [u]
                    while stack:
                        n = stack.pop()
                        if n == h:
                            continue
                        for p in self.pred[n]:
                            if p not in body:
                                body.add(p)
                                stack.append(p)
                    by_header.setdefault(h, set()).update(body)
        loops = [Loop(h, b) for h, b in sorted(by_header.items())]
        for l in loops:
            outer = [c for c in loops if c is not l and l.blocks < c.blocks]
            if outer:
                l.parent = min(outer, key=lambda c: len(c.blocks))
                l.parent.children.append(l)
        self.loops = loops

    def tally(self, blocks):
        t = dict(valu=0, flops=0, trans=0, packed=0, salu=0, smem=0, vmem=0, lds=0, mfma=0, scratch=0)
        for b in blocks:
            for k in self.blocks[b]:
                addr, mnem, ops, tgt = self.ins[k]
                if mnem.startswith("v_mfma"):
                    t["mfma"] += 1
                elif mnem.startswith("v_"):
                    t["valu"] += 1
                    t["flops"] += weight(mnem)
                    t["trans"] += bool(TRANS.match(mnem))
                    t["packed"] += bool(PACKED.match(mnem))
                elif mnem.startswith("s_load") or mnem.startswith("s_buffer_load"):
                    t["smem"] += 1
                elif mnem.startswith("s_"):
                    t["salu"] += 1
                elif mnem.startswith("ds_"):
                    t["lds"] += 1
                elif mnem.startswith("scratch_"):
                    t["scratch"] += 1
                    t["vmem"] += 1
                elif mnem.startswith(("global_", "buffer_", "flat_")):
                    t["vmem"] += 1
        return t

    def depth(self, l):
        d = 0
        while l.parent is not None:
            d, l = d + 1, l.parent
        return d


def pair_model(cfg, mean_series_pairs, frac_odd, W=2, frac_short=0.0, error_map=False):
    """Execution model of gl_pair_kernel / gl_static_kernel in a likelihood / gradient mode, steady state.

    The pixel loop of the CHECK=false instantiation is the first (with `error_map`) or second outermost loop with a substantial
    body; every trip processes W pixels per lane.  Its blocks are weighted as follows:
      * blocks that dominate the loop's latch run once per trip,
      * the inner loop (two series terms per trip) runs `mean_series_pairs` times on average,
      * a conditional block behind the inner loop holding >= 4 packed FMAs is the term the two-term loop leaves over:
        probability `frac_odd` (the caller passes the share of samples whose series length has that parity); its sibling
        branch (register moves only) runs with the complementary probability,
      * the inner loop's preheader runs with probability 1 - `frac_short` (series of fewer than two terms skip the loop),
        the blocks of that bypass with probability `frac_short`,
      * other conditional blocks (optional loads of the mask / error planes: <= 3 VALU, no flops) are counted as executed.
    Returns per-lane flops and VALU wave-instructions per PIXEL, and the decomposition."""
    # likelihood modes compile the steady-state tile twice, in address order: with an error map, without one (gl_pair.hip.h err_tag),
    # then the ragged-end tile; image modes have the one steady-state loop and the ragged one
    hot = _hot_loops(cfg)
    tile = hot[0 if error_map or len(hot) < 3 else 1]
    inner = sorted(tile.children, key=lambda l: l.header)
    inner_blocks = set().union(*[c.blocks for c in inner]) if inner else set()
    own = tile.blocks - inner_blocks
    latches = [b for b in tile.blocks if tile.header in cfg.succ[b]]
    mandatory = set(own)
    for lt in latches:
        mandatory &= cfg.dom[lt]
    mandatory |= {tile.header} & own
    inner_headers = {c.header for c in inner}
    out = {k: float(v) for k, v in cfg.tally(mandatory).items()}
    detail = dict(tile_loop_header=hex(tile.header), n_blocks=len(tile.blocks), pixels_per_lane_per_trip=W,
                  mandatory=cfg.tally(mandatory), inner=[], conditional=[])
    for c in inner:
        t = cfg.tally(c.blocks)
        for k in out:
            out[k] += t[k] * mean_series_pairs
        detail["inner"].append(dict(header=hex(c.header), per_trip=t, trips=mean_series_pairs))
    for b in sorted(own - mandatory):
        t = cfg.tally([b])
        after_loop = any(h in cfg.dom[b] for h in inner_headers)  # only reachable through the series loop
        if after_loop and t["packed"] >= 4:
            prob, why = frac_odd, "series tail: the term left over by the two-term loop"
        elif after_loop and t["valu"] > 3:
            prob, why = 1.0 - frac_odd, "series tail: the other parity (register moves)"
        elif any(h in cfg.succ[b] for h in inner_headers):
            prob, why = 1.0 - frac_short, "series-loop preheader"
        elif t["flops"] == 0 and t["valu"] <= 3:
            prob, why = 1.0, "optional plane load (counted as executed)"
        else:
            prob, why = frac_short, "short-series bypass"
        for k in out:
            out[k] += t[k] * prob
        detail["conditional"].append(dict(block=hex(b), per_trip={k: v for k, v in t.items() if v}, probability=prob, what=why))
    per_pixel = {k: v / W for k, v in out.items()}
    detail["per_pixel"] = per_pixel
    return per_pixel, detail


def _hot_loops(cfg, min_valu=100):
    """Top-level loops with a substantial body, in address order (the CHECK=false steady-state tile loop comes first)."""
    return sorted([l for l in cfg.loops if l.parent is None and cfg.tally(l.blocks)["valu"] >= min_valu], key=lambda l: l.header)


def shp_model(cfg, mean_series_pairs, p_live, W=2, p_lens=1.0):
    """gl_shp_kernel (csrc/gl_shp.hip.h), gradient / likelihood modes.  The steady-state tile loop is the first hot top-level
    loop; per trip (two pixels per lane):
      * blocks that dominate the latch run once,
      * the EPL series loop runs `mean_series_pairs` times,
      * the blocks behind the wave-uniform "any pixel inside the shapelet table" branch -- recognised by what only they hold:
        the table gathers' packed interpolation / contraction (>= 8 packed instructions) or the MFMAs -- run with probability
        `p_live`, the share of wave-tiles the kernel itself counted as live (pad slots of its partial rows),
      * the remaining conditional blocks (optional plane loads, the few moves of the skip path) are counted as executed.
    Round 4 (table mode): wave-tiles provably outside the shapelet table skip the lens forward -- the EPL series loop and the
    conditional blocks on its side of that branch (the lens preamble, which dominates the series loop's header, and what it
    dominates) run with probability `p_lens`, the share of wave-tiles the kernel counted as having run the lens."""
    tile = _hot_loops(cfg)[0]
    inner = sorted(tile.children, key=lambda l: l.header)
    inner_blocks = set().union(*[c.blocks for c in inner]) if inner else set()
    own = tile.blocks - inner_blocks
    latches = [b for b in tile.blocks if tile.header in cfg.succ[b]]
    mandatory = set(own)
    for lt in latches:
        mandatory &= cfg.dom[lt]
    out = {k: float(v) for k, v in cfg.tally(mandatory).items()}
    detail = dict(tile_loop_header=hex(tile.header), pixels_per_lane_per_trip=W, mandatory=cfg.tally(mandatory), inner=[],
                  conditional=[], p_live=p_live)
    for c in inner:
        t = cfg.tally(c.blocks)
        # round 4, table mode: the shapelet chains and the matrix-pipe pass run in ROUNDS over the wave-tile's compacted live
        # pixels (0, 1 or 2 per tile): loops with a large packed body or MFMAs; mean trips = 2 p_live (p_live is handed over as
        # the share of the two chains per lane a tile would run without compaction).  The other inner loop is the EPL series.
        rounds_loop = t["mfma"] > 0 or t["packed"] >= 60
        trips = 2.0 * p_live if rounds_loop else mean_series_pairs * p_lens
        for k in out:
            out[k] += t[k] * trips
        detail["inner"].append(dict(header=hex(c.header), per_trip=t, trips=trips, what="chain / MFMA rounds" if rounds_loop else "EPL series"))
    series_headers = [c.header for c in inner if not (cfg.tally(c.blocks)["mfma"] > 0 or cfg.tally(c.blocks)["packed"] >= 60)]
    # the lens side of the cull branch: the conditional blocks that dominate the series loop (the lens preamble) and everything
    # they dominate up to the join (the series may run zero trips of its loop, so its header does not dominate the lens tail)
    anchors = {b for b in own - mandatory if any(b in cfg.dom[h] for h in series_headers)}
    for b in sorted(own - mandatory):
        t = cfg.tally([b])
        live = t["packed"] >= 8 or t["mfma"] > 0
        lens_side = p_lens < 1.0 and (b in anchors or any(an in cfg.dom[b] for an in anchors))
        prob = p_lens if lens_side else (p_live if live else 1.0)
        for k in out:
            out[k] += t[k] * prob
        if t["valu"] or t["mfma"]:
            detail["conditional"].append(dict(block=hex(b), per_trip={k: v for k, v in t.items() if v}, probability=prob,
                                              what="lens forward of a tile not culled" if lens_side else
                                              ("shapelet chains of a live wave-tile" if live else "counted as executed")))
    per_pixel = {k: v / W for k, v in out.items()}
    detail["per_pixel"] = per_pixel
    return per_pixel, detail


def cluster_model(cfg, W=2, min_valu=1000):
    """gl_cluster_kernel (csrc/gl_cluster.hip.h) at full capacity (every compile-time component slot in use: BASELINE configs 4
    and 5).  The steady-state tile loop is the first hot top-level loop; its component bodies sit behind wave-uniform count
    guards (always taken at capacity).  The only blocks that do not run every trip are the lane-by-lane closed-form NFW
    fallback (X outside the h(X) table or exactly 1: a handful of lanes per launch): the blocks entered as the fall-through of
    an `s_cbranch_execz` (lane-masked code; the count guards are scalar-condition branches), weighted 0.  Checked against
    SQ_INSTS_VALU in profiles/ (C4: 1248 modelled vs 1242-1253 counted per pixel)."""
    tile = _hot_loops(cfg, min_valu)[0]
    latches = [b for b in tile.blocks if tile.header in cfg.succ[b]]
    mandatory = set(tile.blocks)
    for lt in latches:
        mandatory &= cfg.dom[lt]
    out = {k: float(v) for k, v in cfg.tally(mandatory).items()}
    n_fallback = 0
    order = sorted(cfg.blocks)
    nxt = {b: order[i + 1] for i, b in enumerate(order[:-1])}

    def masked_entry(b):
        """b is entered under a lane mask: every way into it is the fall-through of an `s_cbranch_execz` (the lanes that took
        the closed-form NFW branch), as opposed to the wave-uniform count guards (scalar-condition branches)."""
        preds = cfg.pred[b]
        return bool(preds) and all(cfg.ins[cfg.blocks[q][-1]][1] == "s_cbranch_execz" and nxt.get(q) == b for q in preds)

    entries = {b for b in tile.blocks - mandatory if masked_entry(b)}

    def divergent(b):  # inside a lane-masked region: dominated by one of its entries
        return bool(entries & cfg.dom[b])

    for b in sorted(tile.blocks - mandatory):
        t = cfg.tally([b])
        if divergent(b):
            n_fallback += 1
            continue
        for k in out:
            out[k] += t[k]
    per_pixel = {k: v / W for k, v in out.items()}
    detail = dict(tile_loop_header=hex(tile.header), pixels_per_lane_per_trip=W, n_blocks=len(tile.blocks),
                  fallback_blocks_weighted_zero=n_fallback, per_pixel=per_pixel)
    return per_pixel, detail


def execution_model(co, name, md, series, p_live=None, p_lens=None):
    """Dynamic per-pixel counts of a kernel this tool has an execution model for (the specialised pair / static kernels in a
    likelihood or gradient mode), else None.  `series`: dict(mean_pair_trips, frac_odd[, frac_short]) of the batch, or None
    for models without EPL."""
    weights = "v_pk_fma 4, v_pk_mul/add 2, v_fma 2, v_mul/add/sub 1, transcendental 1, other 0 (per lane)"
    ms = re.search(r"gl_shp_kernel<(\d+),", name)
    mc = re.search(r"gl_cluster_kernel<(\d+),", name)
    mw = re.search(r"gl_clusterw_kernel<(\d+),", name)
    if ms or mc or mw:
        if int((ms or mc or mw).group(1)) in (0,):
            return None
        cfg = CFG(disassemble(md["co"], md["symbol"]))
        s_ = series or {}
        if ms:
            per_pixel, detail = shp_model(cfg, float(s_.get("mean_pair_trips", 0.0)), 1.0 if p_live is None else float(p_live),
                                          p_lens=1.0 if p_lens is None else float(p_lens))
        elif mw:
            # gl_clusterw_kernel (csrc/gl_clusterw.hip.h): the four waves of a workgroup walk the SAME 128 pixels of a step (64
            # lanes x one pixel pair each), every wave with a quarter of the components -- a pixel is served by one lane of each
            # of the four waves, so the per-pixel counts are 4 x (one wave's loop body / 2 pixels per lane).  Same block rules as
            # the pixel-split kernel: everything except the lane-masked closed-form NFW fallback runs every step.
            per_lane, detail = cluster_model(cfg, W=2, min_valu=200)
            per_pixel = {k: 4.0 * v for k, v in per_lane.items()}
            detail = dict(detail, waves_per_pixel=4, per_pixel=per_pixel)
        else:
            per_pixel, detail = cluster_model(cfg)
        out = dict(flops_per_pixel=round(per_pixel["flops"], 2), valu_insts_per_pixel=round(per_pixel["valu"], 2),
                   trans_per_pixel=round(per_pixel["trans"], 2), packed_insts_per_pixel=round(per_pixel["packed"], 2),
                   mfma_per_pixel=round(per_pixel["mfma"], 3), flop_weights=weights,
                   model={k: v for k, v in detail.items() if k != "per_pixel"})
        if per_pixel["mfma"]:  # exact-fp32 16x16x4 MFMA: 2 * 16 * 16 * 4 flops per wave-instruction = 32 per lane
            out["mfma_flops_per_pixel"] = round(per_pixel["mfma"] * 32, 2)
        return out
    mi = re.search(r"gl_main_kernel<([13]), 2, (true|false), 1, false>", name)
    if mi and (series or {}).get("n_members"):
        # The interpreter on a model with a galaxy catalogue: the catalogue's member loop -- the innermost
        # single-block loop that loads a member's two constant blocks through scalar loads and carries the tangents (the largest
        # such loop of the gradient kernel: piemd_member_v<v2f, true>, csrc/gl_members.hip.h) -- runs n_members times per pixel
        # pair.  Only that loop is modelled: its share of the launch is what SQ_INSTS_VALU says the rest is (profiles/
        # r4_summary.json::pmc_C6: 16 100 instructions per pixel counted, 14 300 of them this loop at 200 members).
        cfg = CFG(disassemble(md["co"], md["symbol"]))
        loops = [l for l in cfg.loops if not l.children and len(l.blocks) == 1 and cfg.tally(l.blocks)["smem"] >= 4]
        if not loops:
            return None
        body = max((cfg.tally(l.blocks) for l in loops), key=lambda t: t["valu"])
        n = float(series["n_members"])
        return dict(flops_per_pixel=round(body["flops"] * n / 2, 1), valu_insts_per_pixel=round(body["valu"] * n / 2, 1),
                    trans_per_pixel=round(body["trans"] * n / 2, 1), packed_insts_per_pixel=round(body["packed"] * n / 2, 1),
                    flop_weights=weights,
                    model=dict(what="catalogue member loop only (the other components are not modelled)", members=n,
                               member_loop_valu_per_member_and_pixel_pair=body["valu"],
                               member_loop_trans_per_member_and_pixel_pair=body["trans"]))
    m = re.search(r"gl_pair_kernel<(\d+), (float __vector\(2\)|float),", name)
    W = None
    if m:
        W = 2 if "vector" in m.group(2) else 1
    else:
        m = re.search(r"gl_static_kernel<(\d+), (\d+),", name)
        if m:
            W = int(m.group(2))
    if W is None or int(m.group(1)) == 0:  # interpreter kernel / image-only mode: no model here
        return None
    ins = disassemble(md["co"], md["symbol"])
    cfg = CFG(ins)
    s = series or {}
    per_pixel, detail = pair_model(cfg, float(s.get("mean_pair_trips", 0.0)), float(s.get("frac_odd", 0.0)), W,
                                   float(s.get("frac_short", 0.0)), bool(s.get("error_map", False)))
    return dict(flops_per_pixel=round(per_pixel["flops"], 2), valu_insts_per_pixel=round(per_pixel["valu"], 2),
                trans_per_pixel=round(per_pixel["trans"], 2), packed_insts_per_pixel=round(per_pixel["packed"], 2),
                flop_weights="v_pk_fma 4, v_pk_mul/add 2, v_fma 2, v_mul/add/sub 1, transcendental 1, other 0 (per lane)",
                model=dict(tile_loop_header=detail["tile_loop_header"], mandatory=detail["mandatory"],
                           inner=detail["inner"], conditional=detail["conditional"]))


def find_kernel(meta, pattern):
    hits = [k for k in meta if pattern in k]
    if not hits:
        hits = [k for k in meta if re.search(pattern, k)]
    if len(hits) != 1:
        raise SystemExit(f"{len(hits)} kernels match {pattern!r}: {hits[:6]}")
    return hits[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--kernel", help="substring (or regex) of the demangled kernel name")
    ap.add_argument("--series-pairs", type=float, default=0.0, help="mean trips of the two-term EPL series loop")
    ap.add_argument("--frac-odd", type=float, default=0.5, help="fraction of samples with an odd series length")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    co = code_object(args.lib)
    meta = kernel_metadata(co)
    if args.list:
        for name, m in sorted(meta.items(), key=lambda kv: -kv[1]["vgpr_spill_count"]):
            print(f"{m['vgpr_count']:4d} vgpr {m['vgpr_spill_count']:4d} vspill {m['sgpr_spill_count']:4d} sspill "
                  f"{m['scratch_bytes']:5d} B scratch  {name}")
        return
    name = find_kernel(meta, args.kernel)
    ins = disassemble(meta[name]["co"], meta[name]["symbol"])
    cfg = CFG(ins)
    wm = re.search(r"gl_pair_kernel<\d+, float __vector\(2\)", name)
    tm = re.search(r"gl_static_kernel<\d+, (\d+),", name)
    W = 2 if wm else (int(tm.group(1)) if tm else 1)
    per_pixel, detail = pair_model(cfg, args.series_pairs, args.frac_odd, W)
    rep = dict(kernel=name, metadata={k: v for k, v in meta[name].items() if k != "co"}, n_instructions=len(ins), whole_kernel_static=cfg.tally(cfg.blocks),
               loops=[dict(header=hex(l.header), depth=cfg.depth(l), n_blocks=len(l.blocks),
                           own=cfg.tally(l.blocks - set().union(*[c.blocks for c in l.children]) if l.children else l.blocks))
                      for l in cfg.loops],
               model=detail)
    if args.json:
        print(json.dumps(rep))
    else:
        print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
