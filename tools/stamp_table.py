#!/usr/bin/env python3
"""Render the device-timestamp table of a bench.py run (gpurun_out/bench_full.json -> device_timestamps) as text: every launch of
every stream with its [first workgroup in, last workgroup out] interval on the chip's own 100 MHz reference counter
(fz_diag_stamps_*: each workgroup's first wave stores s_memrealtime at entry and, after its stores have been acknowledged, at
exit), then what follows from it -- how many launches were in flight, the fraction of the HBM peak over the span.
usage: python tools/stamp_table.py gpurun_out/bench_full.json > profiles/rNN_device_timestamps.txt"""
import json
import sys


def merged(iv):
    out = []
    for a, b in sorted(iv):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def render(name, sp, kernel):
    if not isinstance(sp, dict) or "table" not in sp:
        print(f"{name}: no table ({sp})")
        return
    t = sp["table"]
    streams = sorted({r["stream"] for r in t})
    print(f"== {name}: {len(streams)} stream(s), {sp['launches_per_stream']} launches each of {kernel}; one replay of the captured steps")
    print(f"   clock: {sp['clock']}")
    print(f"   {'stream':>6} {'launch':>6} {'first wg in us':>15} {'last wg in us':>14} {'last wg out us':>15} {'duration us':>12} {'workgroups':>10}  in flight with")
    rows = sorted(t, key=lambda r: r["start_us"])
    for r in rows:
        others = [f"s{o['stream']}#{o['launch']}" for o in t if o is not r and o["start_us"] < r["end_us"] and o["end_us"] > r["start_us"] and o["stream"] != r["stream"]]
        print(f"   {r['stream']:>6} {r['launch']:>6} {r['start_us']:>15.2f} {r['last_workgroup_in_us']:>14.2f} {r['end_us']:>15.2f} {r['end_us'] - r['start_us']:>12.2f} "
              f"{r['workgroups']:>10}  {' '.join(others) or '-'}")
    iv = [(r["start_us"], r["end_us"]) for r in t]
    busy = sum(b - a for a, b in merged(iv))
    span = max(b for _, b in iv) - min(a for a, _ in iv)
    per = {s: [r for r in t if r["stream"] == s] for s in streams}
    print(f"   span {span:.1f} us, some launch running for {busy:.1f} us of it, sum of durations {sum(b - a for a, b in iv):.1f} us "
          f"-> {sum(b - a for a, b in iv) / busy:.3f} launches in flight on average")
    for s, rs in per.items():
        rs = sorted(rs, key=lambda r: r["launch"])
        full = rs[1:-1]
        gaps = [b["start_us"] - a["end_us"] for a, b in zip(rs, rs[1:])]
        print(f"   stream {s}: full launches {sum(r['end_us'] - r['start_us'] for r in full) / max(1, len(full)):.2f} us on average "
              f"(the first launch is forward-only, the last inverse-only), gap between consecutive launches {sum(gaps) / max(1, len(gaps)):.2f} us, "
              f"dispatcher hands out a launch's workgroups in {sum(r['last_workgroup_in_us'] - r['start_us'] for r in full) / max(1, len(full)):.2f} us")
    nfull = sp["launches_per_stream"] - 1
    total = sp["bytes_per_launch"] * nfull * len(streams)
    print(f"   algorithmic bytes: {len(streams)} x {nfull} full-launch equivalents x {sp['bytes_per_launch']} B = {total} B over {span:.1f} us "
          f"= {total / span / 1e3:.0f} GB/s = {total / span / 1e3 / 8000:.4f} of 8 TB/s   (this round; the median of {len(sp['rounds'])} rounds: {sp['frac']:.4f})")
    print(f"   the same replay by HIP events on each stream: {sp['rounds'][-1]['event_ms']} ms -> {sp['rounds'][-1]['frac_by_events']:.4f}")
    print()


def main():
    full = json.load(open(sys.argv[1]))
    ts = full.get("device_timestamps") or {}
    kernel = full["roofline"]["kernel"]
    print(f"bench.py: value {full['value'] / 1e9:.3f} G NTT/s, {full['config']['streams']} streams, {full['config']['steps_per_launch']} steps per launch; "
          f"roofline.frac {full['roofline']['frac']:.4f} (one stream, HIP events), roofline.chip.frac {full['roofline']['chip']['frac']:.4f} (timed region, HIP events)")
    print("device timestamps: what the KERNELS recorded (no profiler attached, no host clock involved)\n")
    render("timed configuration (roofline.chip.device_clock)", ts.get("chip"), kernel)
    if ts.get("one_stream") is not ts.get("chip"):
        render("one stream (roofline.device_clock)", ts.get("one_stream"), kernel)


if __name__ == "__main__":
    main()
