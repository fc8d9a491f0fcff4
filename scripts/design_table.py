#!/usr/bin/env python3
"""The "Measured" table of DESIGN.md section 3 from a bench details file:
python scripts/design_table.py [bench_details.json] > table.md   (or --write: replaces the block between the
<!-- measured:begin --> / <!-- measured:end --> markers of DESIGN.md).  Every number is the file's own."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def f(v, fmt="{:.2f}"):
    return "-" if v is None or (isinstance(v, float) and v != v) else fmt.format(v)


def table(d: dict, src: str) -> str:
    L = []
    r = d["roofline"]
    B = d["config"]["frames_per_step_per_gpu"]
    L.append(f"Source: `{src}` (`bench.py --steps {d.get('steps')} --warmup {d.get('warmup')}`: {d.get('steps')} timed steps of {B} frames).")
    L.append("")
    L.append("| kernel (per launch / step of 16 384 frames) | ms | algorithmic bytes per unit | GB/s | frac of 8 TB/s | counter traffic / algorithmic | VALU busy | wave cycles waiting |")
    L.append("|---|---|---|---|---|---|---|---|")
    rv = d.get("roofline_valu") or {}
    k1_only = d.get("value_sparse_align_only")
    if k1_only:  # round 6: the headline step is K1 + K4
        L.insert(1, f"Headline step (`value`): K1 + K4 on the matches K2 / K3 produced for the same frames: **{d['value'] / 1e6:.2f} M frames/s** "
                    f"({d['ms_per_step']:.3f} ms per step of {B} frames); K1 on its own: {k1_only / 1e6:.2f} M frames/s.")
    L.append(f"| **K1** `sia_kernel<256>` ({(k1_only or d['value']) / 1e6:.2f} M frames/s on its own) | {f(r['ms'], '{:.3f}')} (last 10: {f(r.get('ms_last_10_launches'), '{:.3f}')}) | "
             f"{r['algorithmic_bytes_per_frame'] / 1e3:.1f} KB / frame | {f(r['achieved'], '{:.0f}')} | **{f(r['frac'], '{:.3f}')}** | {f(r.get('traffic_over_algorithmic'))} | "
             f"{f(rv.get('frac'))} | {f(rv.get('wave_cycles_waiting_frac'))} |")
    f64 = d.get("f64_partials") or {}
    if isinstance(f64.get("roofline"), dict):
        q = f64["roofline"]
        L.append(f"| K1, `-DSIA_F64_PARTIALS` build ({f64['frames_per_s'] / 1e6:.2f} M frames/s) | {f(q['ms'], '{:.3f}')} | {q['algorithmic_bytes_per_frame'] / 1e3:.1f} KB / frame | "
                 f"{f(q['achieved'], '{:.0f}')} | {f(q['frac'], '{:.3f}')} | {f(q.get('traffic_over_algorithmic'))} | - | - |")
    k4 = d.get("roofline_pose_optimize") or {}
    if "ms" in k4:
        L.append(f"| K4 `compose_kernel` + `pose_opt_wave_kernel` (in the headline step) | {f(k4['ms'], '{:.3f}')} | {k4.get('algorithmic_bytes_per_frame', 0) / 1e3:.1f} KB / frame | "
                 f"{f(k4['achieved'], '{:.0f}')} | {f(k4['frac'], '{:.3f}')} | - | - | - |")
    k0 = d.get("k0_pyramid") or {}
    if "ms" in k0:
        L.append(f"| K0 `pyramid_fused_kernel` ({k0['frames']} frames) | {f(k0['ms'], '{:.3f}')} | {k0['algorithmic_bytes_per_frame']} B / frame | {f(k0['achieved'], '{:.0f}')} | "
                 f"**{f(k0['frac'], '{:.3f}')}** | - | - | - |")
    ft = d.get("full_track") or {}
    unit = {"match_prepare": "48 B geometry / trial", "warp": "<= 121 B footprint + 100 B written / trial", "align": "100 + 81 I B / trial",
            "pose_opt_wave": "52 M + 416 B / frame", "seed_prepare": "89 B in + ~90 B workspace / seed", "epi_scan": "64 + 7.13 B / position (union of the windows)",
            "seed_finish": "36 B state + workspace / seed"}
    if "update_seeds/epi_scan_kernel" in (ft.get("kernels") or {}) and "update_seeds/warp_kernel" not in (ft.get("kernels") or {}):
        unit["epi_scan"] = "158 B / warped seed + 96 + 7.13 B / position (union of the windows) + 136 B / aligned seed"  # round 6: with the warp
    for name, v in (ft.get("kernels") or {}).items():
        if not isinstance(v, dict) or (v.get("ms") or 0) < 0.05:
            continue
        key = name.split("/")[-1].replace("_kernel", "")
        valu, lds = v.get("valu") or {}, v.get("lds") or {}
        extra = ""
        if lds.get("bank_conflict_frac_of_port_cycles") is not None and (lds.get("port_busy_frac") or 0) >= 0.005:
            extra = f" (LDS port {f(lds.get('port_busy_frac_vs_busy_cu_cycles', lds.get('port_busy_frac')))} busy, {f(lds['bank_conflict_frac_of_port_cycles'])} of it conflicts)"
        u = unit.get(key, '-')
        if name == "update_seeds/align_kernel" and "update_seeds/seed_finish_kernel" not in (ft.get("kernels") or {}):
            u = "150 + 81 I B / aligned seed + 96 B / seed (seed_finish epilogue)"
        L.append(f"| `{name}` | {f(v['ms'], '{:.3f}')} | {u} | {f(v.get('achieved_GBs'), '{:.0f}')} | {f(v.get('frac'), '{:.3f}')} | "
                 f"{f(v.get('traffic_over_compulsory'))} | {f(valu.get('busy_frac_at_3_cycles_per_instruction'))} | {f(valu.get('wave_cycles_waiting_frac'))}{extra} |")
    if "ms_per_step" in ft:
        st = ft["stages_ms"]
        L.append("")
        L.append(f"Full track (configs[2], representative workload): **{ft['ms_per_step']:.2f} ms per step** = {ft['frames_per_s'] / 1e6:.2f} M frames/s; stages: "
                 + ", ".join(f"{k} {v:.3f}" for k, v in st.items() if v >= 0.02) + " ms.")
    c3 = d.get("config3_xga5_b64") or {}
    if "ms_per_step" in c3:
        L.append(f"configs[3] (1280x960, 5 levels, 1000 patches): B = 64: {c3['ms_per_step']:.3f} ms (latency of one frame, 64 of 256 CUs; frac "
                 f"{c3['roofline']['frac']:.3f}); B = 1024: {c3['frames_per_s_at_batch_1024'] / 1e6:.2f} M frames/s (frac {c3['roofline_at_batch_1024']['frac']:.3f}); the "
                 f"reference's own code on the host: {c3['cpu_baseline']['frames_per_s_1core']:.0f} frames/s on one core.")
    ds = d.get("dropin_sequence") or {}
    if "median_ms_per_frame_hip_dropin" in ds:
        L.append(f"Single stream (drop-in, 752x480, 600 frames, map of the reference trace's size): **{ds['median_ms_per_frame_hip_dropin']['tot_time']:.3f} ms** per frame "
                 f"({ds['median_ms_per_frame_hip_dropin_deferred_mapper']['tot_time']:.3f} with the deferred mapper) against {ds['median_ms_per_frame_cpu_reference']['tot_time']:.3f} ms for the "
                 f"all-CPU reference on the same host"
                 + (f"; in a process of its own: {ds['median_ms_per_frame_in_a_process_of_its_own']['hip_dropin']:.3f} / "
                    f"{ds['median_ms_per_frame_in_a_process_of_its_own']['hip_dropin_deferred_mapper']:.3f} ms"
                    if isinstance(ds.get("median_ms_per_frame_in_a_process_of_its_own"), dict) and "hip_dropin" in ds["median_ms_per_frame_in_a_process_of_its_own"] else "")
                 + (f", bound to the GPU's NUMA node: {ds['median_ms_per_frame_in_a_process_of_its_own']['hip_dropin_bound_to_the_gpus_numa_node']:.3f} / "
                    f"{ds['median_ms_per_frame_in_a_process_of_its_own']['hip_dropin_deferred_mapper_bound_to_the_gpus_numa_node']:.3f} ms"
                    if isinstance(ds.get("median_ms_per_frame_in_a_process_of_its_own"), dict)
                    and "hip_dropin_deferred_mapper_bound_to_the_gpus_numa_node" in ds["median_ms_per_frame_in_a_process_of_its_own"] else "")
                 + (f"; with `DepthFilter`'s own thread (the reference's default): {ds['median_ms_per_frame_mapper_thread']['hip_dropin']:.3f} against "
                    f"{ds['median_ms_per_frame_mapper_thread']['cpu_reference']:.3f} ms" if isinstance(ds.get("median_ms_per_frame_mapper_thread"), dict) else "") + ".")
    rc = d.get("reference_cameras") or {}
    if isinstance(rc.get("cameras"), dict):
        L.append("")
        L.append(f"The reference's own cameras (`svo_ros/param/camera_pinhole.yaml` = radtan, `camera_atan.yaml`; 752x480, {rc.get('frames_per_step')} frames per step, "
                 "configs[1]'s shape; `pinhole_undistorted` = camera_pinhole.yaml without its distortion):")
        L.append("")
        L.append("| camera | K1 M frames/s | frac | traffic / algorithmic | Gauss-Newton iterations per frame (reference TU on a sample) | K1 time per iteration vs undistorted | full track ms per step (update_seeds) | drop-in ms per frame |")
        L.append("|---|---|---|---|---|---|---|---|")
        for name, r in rc["cameras"].items():
            sa, ftc, dr = r.get("sparse_align") or {}, r.get("full_track") or {}, r.get("dropin") or {}
            par = sa.get("parity") or {}
            L.append(f"| {name} | {f((sa.get('frames_per_s') or 0) / 1e6)} | {f((sa.get('roofline') or {}).get('frac'), '{:.3f}')} | {f((sa.get('roofline') or {}).get('traffic_over_algorithmic'))} | "
                     f"{f(sa.get('mean_gn_iterations_per_frame'), '{:.1f}')} ({f(par.get('mean_gn_iterations_per_frame_reference'), '{:.1f}')}; same counts {f(par.get('same_iteration_counts_frac'), '{:.3f}')}, "
                     f"log-norm max {f(par.get('se3_lognorm_max'), '{:.1e}')}) | {f(sa.get('per_iteration_over_undistorted_pinhole'), '{:.3f}')} | "
                     f"{f(ftc.get('ms_per_step'), '{:.2f}')} ({f((ftc.get('stages_ms') or {}).get('update_seeds'), '{:.2f}')}) | {f(dr.get('tot_time'), '{:.3f}')} |")
        L.append("")
    cb = d.get("cpu_baseline") or {}
    if "value" in cb:
        L.append(f"CPU baseline (the reference's own `sparse_img_align.cpp`, {cb.get('cpu_model', 'host')}): {cb['value']:.0f} frames/s on one core in the bit-comparable build, "
                 f"{f(cb.get('value_release_flags'), '{:.0f}')} with the reference's release flags; best thread count ({cb.get('best_threads')}): {cb['value_best_threads']:.0f} / "
                 f"{f(cb.get('value_release_flags_best_threads'), '{:.0f}')} frames/s.")
    par = d.get("parity") or {}
    if par:
        L.append(f"Parity of the headline run against that translation unit over {par['frames_compared']} frames: SE(3) log-norm max {par['se3_lognorm_max']:.2e}, median "
                 f"{par['se3_lognorm_median']:.1e}; identical per-level iteration counts {par['same_iteration_counts_frac']:.4f}.")
    import textwrap
    out, in_table = [], False
    for l in L:
        if l.startswith("|"):
            out.append(l)
            in_table = True
            continue
        if l and out and out[-1] != "" and not in_table and not l.startswith("Source:"):
            out.append("")  # paragraphs of their own
        in_table = False
        out.extend(textwrap.wrap(l, 108, break_long_words=False, break_on_hyphens=False) if len(l) > 110 else [l])
    return "\n".join(out)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    src = args[0] if args else "bench_details.json"
    d = json.load(open(src))
    t = table(d, os.path.relpath(src, ROOT) if os.path.isabs(src) else src)
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        b, e = "<!-- measured:begin -->", "<!-- measured:end -->"
        i, j = s.index(b) + len(b), s.index(e)
        open(p, "w").write(s[:i] + "\n" + t + "\n" + s[j:])
    else:
        print(t)


if __name__ == "__main__":
    main()
