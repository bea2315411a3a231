"""Turn the output directory of tools/profile_round.sh into the committed summaries under profiles/.
usage: python tools/make_profiles.py gpurun_out/<dir> r03"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLK = 2.4e9
SIMDS = 1024


def last_json(path):
    return json.loads(open(path).read().strip().split("\n")[-1])


def bench_lines(src, tag, P, b, d):
    t = json.load(open(f"{src}/pmc_traffic/cconv_hbm_traffic.json"))
    json.dump(t, open(f"{P}/{tag}_cconv_hbm_traffic.json", "w"), indent=1)
    tb = t["by_kernel"]
    lines = [f"# Bench lines ({tag}, one MI355X, `gpurun`)\n",
             "`python bench.py --steps 5 --warmup 3` (steps 4-8 of the rollout):\n", "```json", json.dumps(b), "```\n",
             "`python bench.py --steps 20 --warmup 5 --cpu-side 0` (the driver's window, steps 6-25: the scene degrades while it runs -- particles "
             "leak through the shell from step ~10 on, rows get longer; `scene_state` in the line says by how much):\n", "```json", json.dumps(d), "```\n",
             "Per kernel (ms per step, fraction of 8 TB/s by the contract's algorithmic bytes, HBM bytes per launch from the PMC passes of "
             f"profiles/{tag}_cconv_hbm_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE), steps 4-8 | the driver's window:\n",
             "| kernel | ms/step | frac | ms/step (driver) | frac (driver) | algorithmic GB / launch | PMC HBM GB / launch |", "|---|---:|---:|---:|---:|---:|---:|"]
    dk = d["roofline_groups"]["by_kernel"]
    for k, v in b["roofline_groups"]["by_kernel"].items():
        hb = [x for n, x in tb.items() if n.endswith("::" + k) or (k == "lat_conv_kernel" and "lat_conv" in n)
              or (k.startswith("cconv_sct_kernel<") and n.split("::")[-1].startswith(k[:-1]))]  # (r06's first lines say <4>, rocprof <4, 16>)
        pm = f"{sum(x['hbm_bytes_per_launch'] * x['launches'] for x in hb) / max(sum(x['launches'] for x in hb), 1) / 1e9:.2f}" if hb else ""
        w = dk.get(k, dict(ms_per_step=float('nan'), frac=float('nan')))
        lines.append(f"| `{k}` | {v['ms_per_step']:.2f} | {v['frac']:.3f} | {w['ms_per_step']:.2f} | {w['frac']:.3f} | {v['algorithmic_bytes_per_launch'] / 1e9:.2f} | {pm} |")
    for name, x in (("steps 4-8", b), ("driver's window", d)):
        g = x["roofline_groups"]["neighbour_list"]
        lines.append(f"\n{name}: {x['ms_per_step']:.2f} ms per step = {x['value']:.4g} particle-steps/s; all neighbour-list kernels {g['ms_per_step']:.2f} ms per step at "
                     f"{g['frac']:.3f} of the roofline; dominant `{x['roofline']['kernel']}` {x['roofline']['frac']:.3f}; searches {x['search']['ms_per_step']:.2f} ms "
                     f"({x['search']['achieved']:.0f} GB/s on their own algorithmic bytes); step - sum of the dmcf launches = "
                     f"{x['ms_per_step'] - sum(x['kernel_ms_per_step'].values()):.2f} ms.")
    fr = [(n, x) for n, x in tb.items() if "frs_" in n]
    if fr:
        lines.append("\nSearch kernels (PMC HBM GB per launch): " + ", ".join(f"`{n.split('::')[-1]}` {x['hbm_bytes_per_launch'] / 1e9:.3f}" for n, x in fr))
    open(f"{P}/{tag}_bench_lines.md", "w").write("\n".join(lines) + "\n")


def kernel_stats(src, tag, P, b):
    stats = open(f"{src}/kernel_stats.md").read()
    total = [l for l in stats.splitlines() if l.startswith("total kernel time")][0]
    dom = b["roofline"]["kernel"].split("::")[-1]
    row = [l for l in stats.splitlines() if dom in l][0].split("|")
    open(f"{P}/{tag}_bench_1m_kernel_stats.md", "w").write(
        f"rocprofv3 --kernel-trace --stats of `python bench.py --steps 5 --warmup 3 --cpu-side 0` (8 rollout steps of the 1M-particle box, one "
        f"MI355X), summarised by tools/rocpd_stats.py.  The untraced run of the same command: {b['ms_per_step']:.1f} ms per step; {total}: "
        f"the GPU is never idle.  Dominant neighbour-list kernel `{dom}` (splat F on the 3e8-pair 24-channel layers L3, L4): "
        f"{float(row[4]) / 1e3:.2f} ms per launch here, {b['roofline']['avg_launch_ms']:.2f} ms by bench.py's HIP events.\n\n" + stats)


def pmc(src, tag, P):
    """SQ counters -> derived per-launch figures (SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles, SQ_VALU_MFMA_BUSY
    cycles; sums over the launches of the command)."""
    mb = {}
    if os.path.exists(f"{src}/microbench.log"):
        for l in open(f"{src}/microbench.log"):
            if ": pairs" in l:
                name = l.split()[0]
                mb[name] = (float(l.split("pairs")[1].split("M")[0]) * 1e6, float(l.split("M")[1].split("ms")[0]))
    case_of = dict(sct_S4="S4", pair_L4="L4", pair_L3="L3", pair4_LQ="LQ", cls1_L8="L8", cls2_L5="L5", cls2n_L6="L6", cls4_LP="LP", z3_L14="L14", direct_ASCC="ASCC",
                   ws_L14="L14", ws_L2="L2")
    # (forced kernels: their time is in microbench_short.log, not in the default-dispatch table)
    forced = {}
    if os.path.exists(f"{src}/microbench_short.log"):
        k = None
        for l in open(f"{src}/microbench_short.log"):
            if l.startswith("== DMCF_CCONV_KERNEL="):
                k = l.strip().split("=")[-1]
            elif ": pairs" in l and k:
                forced[(k, l.split()[0])] = (float(l.split("pairs")[1].split("M")[0]) * 1e6, float(l.split("M")[1].split("ms")[0]))
    lines = [f"# SQ counters of every kernel above 3 % of the 1M step ({tag})\n",
             "`tools/pmc_kernel.sh`: three separate `rocprofv3 --kernel-trace --pmc` passes per kernel (no other trace domain), on the micro-benchmark "
             "case that exercises the kernel (`tools/microbench.py`, 1 + 5 launches; the search: `tools/bench_search.py`, the lattice form: "
            "`tools/bench_lattice.py`, pairs and ms averaged over the lists / layers the kernel ran on).  Derived per launch: "
             "SQ_INSTS_* / launches / pairs = wave instructions per neighbour pair; busy = SQ_ACTIVE_INST_VALU x 4 (quad-cycles) resp. "
             "SQ_VALU_MFMA_BUSY_CYCLES over the SIMD-cycles of a launch = 1024 SIMDs x the launch's OWN clock cycles: GRBM_GUI_ACTIVE of a "
             "fourth pass, per XCD (round 4 assumed 2.4 GHz x the launch time and read 110 % for the search: the part clocks lower under "
             "that kernel; the column `GHz` is cycles / time -- and the search STILL reads 113 % at its measured 2.34 GHz: SQ_ACTIVE_INST_VALU "
             "sums, over the waves of a SIMD, the cycles each wave has a vector instruction in flight, and with 7 waves per SIMD those overlap "
             "in the pipeline.  So `VALU active` is an occupancy of the vector pipe by waves, not a utilisation: it bounds nothing above "
             "~2 waves per SIMD; the instruction counts per pair are the figures to go by there).  For the splat kernels (2 - 4 waves per SIMD) "
             "VALU active + matrix busy is what the verdict asks for: the two pipes of a SIMD do not overlap on this part (DESIGN section "
             "4.2), so their sum is the fraction of SIMD time that issues arithmetic.\n",
             "| kernel | case | pairs | ms | GHz | VALU / pair | SALU / pair | LDS / pair | VMEM / pair | matrix instr / pair | VALU active | matrix busy | VALU + matrix | waves per SIMD | LDS bank-conflict cycles / LDS cycles |",
             "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    raw = {}
    # the search (tools/bench_search.py: 1 + 5 padded launches per list) and the lattice form (tools/bench_lattice.py: 1 + 5
    # launches per layer): pairs / ms per launch averaged over the lists / layers the kernel ran on
    special = {}
    if os.path.exists(f"{src}/search.log"):
        rows = [(float(l.split("M pairs")[0].split(":")[1]) * 1e6, float(l.split("padded")[1].split("ms")[0])) for l in open(f"{src}/search.log") if "padded" in l]
        if rows:
            special["frs_query_padded"] = ("5 lists + s2->s2", sum(r[0] for r in rows) / len(rows), sum(r[1] for r in rows) / len(rows))
    if os.path.exists(f"{src}/lattice.log"):
        rows = [(float(l.split("pairs")[1].split("M")[0]) * 1e6, float(l.split("lattice")[1].split("ms")[0])) for l in open(f"{src}/lattice.log") if "lattice" in l and "pairs" in l]
        if len(rows) == 3:
            special["lat_conv_kernel<1, 2>"] = ("s1->s1 8->16, s1->s2 8->8 (stencil offsets x outputs as pairs)", (rows[0][0] + rows[1][0]) / 2, (rows[0][1] + rows[1][1]) / 2)
            special["lat_conv_kernel<1, 1>"] = ("s2->s2 4->8", rows[2][0], rows[2][1])
    for f in sorted(glob.glob(f"{src}/pmc/*.json")):
        label = os.path.basename(f)[:-5]
        res = json.load(open(f))
        raw[label] = res
        for k, v in res.items():
            n = max(v.get("launches", 1), 1)
            case = case_of.get(label)
            if case and (label.split("_")[0], case) in forced:
                pairs, ms = forced[(label.split("_")[0], case)]
            elif case and case in mb:
                pairs, ms = mb[case]
            elif k.split("::")[-1] in special:
                case, pairs, ms = special[k.split("::")[-1]]
            else:
                pairs, ms = None, None
            if pairs:
                # the launch's own clock: GRBM_GUI_ACTIVE is summed over the 8 XCDs (a figure outside 1 .. 3 GHz means it is
                # not: then the counter is taken as it is; without the counter 2.4 GHz is assumed and said so)
                gui = v.get("GRBM_GUI_ACTIVE", 0.0) / n
                clk = None
                for div in (8.0, 1.0, 32.0):
                    c = gui / div / (ms * 1e-3)
                    if 1.0e9 <= c <= 3.0e9:
                        clk = c
                        break
                ghz = f"{clk / 1e9:.2f}" if clk else "2.4 (assumed)"
                cyc = SIMDS * (clk or CLK) * ms * 1e-3
                per = lambda c: v.get(c, 0.0) / n / pairs
                valu, mat = 100 * v.get('SQ_ACTIVE_INST_VALU', 0) * 4 / n / cyc, 100 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / n / cyc
                lines.append(f"| `{k.split('::')[-1]}` | {case} | {pairs / 1e6:.1f}M | {ms:.2f} | {ghz} | {per('SQ_INSTS_VALU'):.2f} | {per('SQ_INSTS_SALU'):.2f} | "
                             f"{per('SQ_INSTS_LDS'):.2f} | {per('SQ_INSTS_VMEM'):.3f} | {per('SQ_INSTS_MFMA'):.2f} | "
                             f"{valu:.0f} % | {mat:.0f} % | {valu + mat:.0f} % | "
                             f"{v.get('SQ_WAVE_CYCLES', 0) * 4 / n / cyc:.2f} | {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.2f} |")
            else:
                lines.append(f"| `{k.split('::')[-1]}` | {label} | | | | | | | | | | | | | {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.2f} |")
    lines.append("\nRaw sums:\n\n```json\n" + json.dumps(raw, indent=1) + "\n```")
    open(f"{P}/{tag}_kernel_pmc.md", "w").write("\n".join(lines) + "\n")


def rollouts(src, tag, P):
    rows = [f"# Full-length rollouts (tools/long_rollout.py, one MI355X, {tag})\n",
            "README.md:79 of the reference names 200 (Liquid3d) / 600 (WaterRamps) / 3200 (WBC-SPH) frames.  Config 4: the 100,000-particle dam "
            "break as specified (h = 0.05, jitter seed 0, open 2-layer tank), Liquid3d weights; configs 2 / 3: the architectures with seeded "
            "stand-in weights (their checkpoints are not shipped) on ~2k / 3.6k-particle 2-D boxes.  Every step: finite, momentum residual = "
            "|sum of the ASCC output over fluid + boundary| / sum of |.|; the first 5 steps against the CPU oracle fed with the HIP path's own state.  "
            "The searches return the set of the distance test (the default: symmetric lists).  The loop runs as Simulator.run_rollout does "
            "(pipelines.simulator.steady_steps: Python's cyclic GC off).\n",
            "| rollout | particles (+boundary) | steps | all finite | worst momentum residual | worst oracle rel err (5 steps) | repeated steps | "
            "steps with a fresh device allocation (after step 3) | median ms/step | p99 ms | max ms | reserved GiB at the end |",
            "|---|---|---:|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for n in ("liquid3d_dam", "waterramps", "wbcsph"):
        if not os.path.exists(f"{src}/rollout_{n}.json"):
            continue
        s = json.load(open(f"{src}/rollout_{n}.json"))["summary"]
        rows.append(f"| {n} | {s['particles']} (+{s['boundary']}) | {s['steps']} | {s['all_finite']} | {s['worst_momentum_residual']:.2e} | "
                    f"{s['worst_oracle_rel_err']:.2e} | {s['repeated_steps']} | {s['steps_with_device_alloc']} | {s['ms_median']:.2f} | "
                    f"{s['ms_p99']:.1f} | {s['ms_max_after_step3']:.1f} | {s['reserved_gib_last']:.2f} |")
    open(f"{P}/{tag}_long_rollouts.md", "w").write("\n".join(rows) + "\n")


def microbench(src, tag, P):
    if os.path.exists(f"{src}/microbench.log"):
        body = "".join(l for l in open(f"{src}/microbench.log") if ": pairs" in l)
        open(f"{P}/{tag}_microbench.md", "w").write(
            f"# Kernel micro-benchmarks on the 1M box ({tag}, `python tools/microbench.py`, default dispatch with the model's row_length_hint)\n\n```\n{body}```\n")


def ghosts(src, tag, P):
    out = [f"# Ghost fractions of the 2x2x2 block decomposition ({tag}, virtual ranks on one MI355X, `tools/ghost_fraction.py`)\n"]
    for name, title in (("ghost_weak", "weak scaling: 100^3 particles per rank (one box of 200^3), `bench.py --gpus 8`"),
                        ("ghost_strong", "strong scaling: ONE box of 100^3 particles, 50^3 per rank, `bench.py --gpus 8 --scaling strong`")):
        f = f"{src}/{name}.json"
        if not os.path.exists(f):
            continue
        d = json.loads(open(f).read().strip().split("\n")[0])
        out += [f"## {title}\n", "| rank | block | s0 owned | s0 ghosts (widest) | s1 owned | s1 ghosts | s2 owned | s2 ghosts | feature rows / step |", "|---|---|---:|---:|---:|---:|---:|---:|---:|"]
        for r in d["ranks"]:
            s = r["sets"]
            cell = lambda k: (f"{s[k]['owned']} | {s[k]['ghosts_widest']} ({100.0 * s[k]['ghosts_widest'] / max(s[k]['owned'], 1):.1f} %)") if k in s else " | "
            out.append(f"| {r['rank']} | {tuple(r['block'])} | {cell('s0')} | {cell('s1')} | {cell('s2')} | {r['feature_rows_per_step']} |")
        last = max(r["step_seconds"][-1] for r in d["ranks"])
        out.append(f"\nlast step: {1e3 * last:.1f} ms for all {len(d['ranks'])} ranks on ONE GPU = {1e3 * last / len(d['ranks']):.1f} ms of GPU time per rank and step.\n")
    if len(out) > 1:
        open(f"{P}/{tag}_ghost_fraction.md", "w").write("\n".join(out) + "\n")


def wave_specialisation(src, tag, P):
    if not os.path.exists(f"{src}/microbench_short.log"):
        return
    out = [f"# Wave specialisation on the short-row layers ({tag}; `cconv_ws.hip`, DESIGN section 4.2 splat H)\n",
           "The four layers at the network's base radius (s0 -> s0, R = 0.1, ~33 pairs per row, 1.12M outputs) under every kernel that serves "
           "them (`DMCF_CCONV_KERNEL=... ONLY=L14,L2,L5,IN python tools/microbench.py`; ws = producers / consumers in a persistent "
           "workgroup, z3 = splat E, cls = splat D, pair = splat F with one set of waves):\n", "```", open(f"{src}/microbench_short.log").read().rstrip(), "```\n"]
    if os.path.exists(f"{src}/wtrace.log"):
        out += ["Where a producer's and a consumer's clocks go (`make -C dmcf_amd/csrc ws_trace`, `tools/wtrace.py`: cycle stamps at the phase "
                "boundaries, first producer / consumer of every 16th workgroup; the stamps cost ~5 %):\n", "```", open(f"{src}/wtrace.log").read().rstrip(), "```"]
    open(f"{P}/{tag}_wave_specialisation.md", "w").write("\n".join(out) + "\n")


def small_configs(src, tag, P):
    out = [f"# The small configurations: where a step's time goes ({tag}, `tools/profile_small.py <rollout> 100` after 20 warm-up steps)\n",
           "BASELINE.json configs 2 / 3 / 4 are paced by the HOST, not the GPU: per step the eager time, the synchronising calls torch "
           "reports, the host profile (cProfile, tottime) and -- from `rocprofv3 --kernel-trace --stats` of the same command (220 steps "
           "+ set-up) -- the kernels.\n"]
    for r in ("waterramps", "wbcsph", "liquid3d_dam"):
        if not os.path.exists(f"{src}/small_{r}.log"):
            continue
        log = [l.rstrip()[:150] for l in open(f"{src}/small_{r}.log") if "amdgpu.ids" not in l and "UserWarning" not in l and "_cuda_set_sync" not in l]
        out += [f"## {r}\n", "```", "\n".join(log[:45]), "```\n"]
        ks = f"{src}/small_kernel_stats_{r}.md"
        if os.path.exists(ks):
            t = open(ks).read().splitlines()
            out += ["\n".join(t[:22]), "", t[-1], ""]
    open(f"{P}/{tag}_small_configs.md", "w").write("\n".join(out) + "\n")


def virtual_ranks(src, tag, P):
    def row(prefix, grid, blocks):
        n = grid.replace(" ", "")
        st, lg = f"{src}/{prefix}_stats_{n}.md", f"{src}/{prefix}_{n}.log"
        if not (os.path.exists(st) and os.path.exists(lg)):
            return None
        tot = [l for l in open(st) if l.startswith("total kernel time")][0].split()
        ms, disp = float(tot[3]), int(tot[6])
        ranks = eval(grid.replace(" ", "*"))
        steps = [l for l in open(lg) if l.startswith("steps (all")]
        wall = steps[0].split("per rank:")[1].split(";")[0].strip() if steps else ""
        trips = steps[0].split("host round trips per step and rank:")[1].strip() if steps else ""
        return f"| {blocks} | {ranks} | {ms / ranks:.1f} | {ms / ranks / 4:.1f} | {disp / ranks / 4:.0f} | {wall} | {trips} |"
    head = ["| blocks | ranks | kernel ms per rank (4 steps) | per step | dispatches per rank and step | wall ms per rank, steps 1 - 4 | host round trips per rank and step (this communicator) |",
            "|---|---:|---:|---:|---:|---|---:|"]
    out = [f"# Kernel time and dispatches per virtual rank ({tag}, `tools/ghost_fraction.py 100 gx gy gz 4 [strong]` under `rocprofv3 --kernel-trace --stats`)\n",
           "N virtual ranks = N threads of ONE process on ONE MI355X, each with the real kernels on its own block (LocalComm instead of RCCL: every "
           "exchange is a device copy).  The ranks' kernels serialise on the one device, so the SUM of kernel durations / N is what a rank's step "
           "costs its GPU -- ghost rows, plan building and exchange copies included.  Four steps (the first one with the exact searches).\n",
           "## weak scaling: 100^3 particles PER rank\n"] + head
    for g, b in (("1 1 1", "1x1x1"), ("2 1 1", "2x1x1"), ("2 2 1", "2x2x1"), ("2 2 2", "2x2x2")):
        r = row("vranks", g, b)
        if r:
            out.append(r)
    out += ["\n## strong scaling: ONE box of 100^3 particles split over the ranks (BASELINE.json's \"1M particles @ 1/2/4/8 GPUs\" read literally)\n"] + head
    for g, b in (("1 1 1", "1x1x1"), ("2 1 1", "2x1x1 (50 x 100 x 100 per rank)"), ("2 2 1", "2x2x1 (50 x 50 x 100)"), ("2 2 2", "2x2x2 (50^3)")):
        r = row("vranks" if g == "1 1 1" else "sranks", g, b)
        if r:
            out.append(r)
    open(f"{P}/{tag}_virtual_rank_kernel_time.md", "w").write("\n".join(out) + "\n")


def scatter(src, tag, P):
    if not os.path.exists(f"{src}/bench_scatter.log"):
        return
    clean = lambda t: "".join(l for l in t.splitlines(True) if "amdgpu.ids" not in l).rstrip()
    out = [f"# Splat S (filter first, input stationary, fixed-point sums) against the gather kernels ({tag}; `cconv_sct.hip`, DESIGN section 4.3)\n",
           "`python tools/bench_scatter.py`: the particles of the 100^3 box + shell onto the coarse `grid_pos` lattices, whole calls (memset + bound "
           "+ kernel + finish for S), min of 5; `plan` = dmcf_cconv_scatter_plan, once per pair of point sets and step:\n", "```",
           clean(open(f"{src}/bench_scatter.log").read()), "```\n"]
    if os.path.exists(f"{src}/sct_variants.log"):
        out += ["The same with one part of the kernel removed each (`make -C dmcf_amd/csrc sct_variants`, `tools/sct_variants.sh`; wrong results, "
                "right timing of what is left): NOATOM = no LDS adds (and therefore nothing to flush), NOFLUSH = no global atomics, NOGATHER = "
                "G_j read at one fixed cell:\n", "```", clean(open(f"{src}/sct_variants.log").read()), "```"]
    open(f"{P}/{tag}_scatter.md", "w").write("\n".join(out) + "\n")


def more_bench_lines(src, tag, P):
    """Appends the lines of the other BASELINE configs and of the settled scene to <tag>_bench_lines.md."""
    path = f"{P}/{tag}_bench_lines.md"
    out = [open(path).read().rstrip(), ""]
    for name, title in (("bench_settled", "`python bench.py --scene settled --steps 20 --warmup 5 --cpu-side 0` (every timed step starts from the state after the warm-up)"),
                        ("bench_waterramps", "`python bench.py --config waterramps --steps 600 --warmup 20` (BASELINE config 2)"),
                        ("bench_wbcsph", "`python bench.py --config wbcsph --steps 600 --warmup 20` (BASELINE config 3)"),
                        ("bench_liquid3d_dam_first20", "`python bench.py --config liquid3d_dam --steps 20 --warmup 1` (BASELINE config 4, steps 2 - 21)"),
                        ("bench_liquid3d_dam", "`python bench.py --config liquid3d_dam --steps 200 --warmup 20 --cpu-side 0` (BASELINE config 4, steps 21 - 220: the scene dissolves)")):
        f = f"{src}/{name}.log"
        if not os.path.exists(f):
            continue
        try:
            d = last_json(f)
        except ValueError:
            out += [f"{title}: FAILED", "```", open(f).read()[-1500:], "```", ""]
            continue
        keep = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "step_ms",
                                      "dispatches_per_step", "synchronising_reads_per_step", "library_launch_ms_per_step", "step_ms_instrumented",
                                      "kernel_ms_per_step", "scene_state", "cpu_baseline") if k in d}
        bk = {k: dict(ms_per_step=round(v["ms_per_step"], 4), launches=v["launches"], frac=round(v["frac"], 4), frac_flops=round(v.get("frac_flops", 0) or 0, 4))
              for k, v in d.get("roofline_groups", {}).get("by_kernel", {}).items()}
        out += [title + ":\n", "```json", json.dumps(keep), "```", "per kernel: `" + json.dumps(bk) + "`", ""]
    open(path, "w").write("\n".join(out) + "\n")


def main(src, tag):
    P = os.path.join(ROOT, "profiles")
    b, d = last_json(f"{src}/bench.log"), last_json(f"{src}/bench_driver.log")
    bench_lines(src, tag, P, b, d)
    kernel_stats(src, tag, P, b)
    pmc(src, tag, P)
    rollouts(src, tag, P)
    microbench(src, tag, P)
    ghosts(src, tag, P)
    wave_specialisation(src, tag, P)
    small_configs(src, tag, P)
    if os.path.exists(f"{src}/vranks_111.log"):
        virtual_ranks(src, tag, P)
    scatter(src, tag, P)
    more_bench_lines(src, tag, P)
    print("wrote", [f for f in sorted(os.listdir(P)) if f.startswith(tag)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
