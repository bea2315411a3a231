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
        hb = [x for n, x in tb.items() if n.endswith("::" + k) or (k == "lat_conv_kernel" and "lat_conv" in n)]
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
    case_of = dict(pair_L4="L4", pair_L3="L3", pair4_LQ="LQ", cls1_L8="L8", cls2_L5="L5", cls2n_L6="L6", cls4_LP="LP", z3_L14="L14", direct_ASCC="ASCC")
    lines = [f"# SQ counters of every kernel above 3 % of the 1M step ({tag})\n",
             "`tools/pmc_kernel.sh`: three separate `rocprofv3 --kernel-trace --pmc` passes per kernel (no other trace domain), on the micro-benchmark "
             "case that exercises the kernel (`tools/microbench.py`, 1 + 5 launches; the search: `tools/bench_search.py`, the lattice form: "
            "`tools/bench_lattice.py`, pairs and ms averaged over the lists / layers the kernel ran on).  Derived per launch: "
             "SQ_INSTS_* / launches / pairs = wave instructions per neighbour pair; busy = SQ_ACTIVE_INST_VALU x 4 (quad-cycles) resp. "
             "SQ_VALU_MFMA_BUSY_CYCLES over the SIMD-cycles of a launch (1024 SIMDs x 2.4 GHz x its time).\n",
             "| kernel | case | pairs | ms | VALU / pair | SALU / pair | LDS / pair | VMEM / pair | matrix instr / pair | VALU busy | matrix busy | waves per SIMD | LDS bank-conflict cycles / LDS cycles |",
             "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
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
            if case and case in mb:
                pairs, ms = mb[case]
            elif k.split("::")[-1] in special:
                case, pairs, ms = special[k.split("::")[-1]]
            else:
                pairs, ms = None, None
            if pairs:
                cyc = SIMDS * CLK * ms * 1e-3
                per = lambda c: v.get(c, 0.0) / n / pairs
                lines.append(f"| `{k.split('::')[-1]}` | {case} | {pairs / 1e6:.1f}M | {ms:.2f} | {per('SQ_INSTS_VALU'):.2f} | {per('SQ_INSTS_SALU'):.2f} | "
                             f"{per('SQ_INSTS_LDS'):.2f} | {per('SQ_INSTS_VMEM'):.3f} | {per('SQ_INSTS_MFMA'):.2f} | "
                             f"{100 * v.get('SQ_ACTIVE_INST_VALU', 0) * 4 / n / cyc:.0f} % | {100 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / n / cyc:.0f} % | "
                             f"{v.get('SQ_WAVE_CYCLES', 0) * 4 / n / cyc:.2f} | {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.2f} |")
            else:
                lines.append(f"| `{k.split('::')[-1]}` | {label} | | | | | | | | | | | {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.2f} |")
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


def main(src, tag):
    P = os.path.join(ROOT, "profiles")
    b, d = last_json(f"{src}/bench.log"), last_json(f"{src}/bench_driver.log")
    bench_lines(src, tag, P, b, d)
    kernel_stats(src, tag, P, b)
    pmc(src, tag, P)
    rollouts(src, tag, P)
    microbench(src, tag, P)
    ghosts(src, tag, P)
    print("wrote", [f for f in sorted(os.listdir(P)) if f.startswith(tag)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
