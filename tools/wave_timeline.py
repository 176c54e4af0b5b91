#!/usr/bin/env python3
"""Diagnostic: per-wave timeline of one k_render_skip launch (rt_debug_wave_trace, csrc/rt_debug.h).  Every wave records its start and
end on the 100 MHz clock plus HW_ID / XCC_ID; this prints how the launch's time is made up: when the last wave was
dispatched, how long the longest waves ran, how busy the SIMDs were over time.
usage: wave_timeline.py [w h spp level [coop_thr [coop_level [narrow_max [narrow_l2]]]]]      """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta


def main():
    w, h, spp, level = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1920, 1080, 1, 8)
    if len(sys.argv) > 5:
        rta.capi.debug_set(rta.capi.DEBUG_COOP_THR, int(sys.argv[5]))
    if len(sys.argv) > 6:
        rta.capi.debug_set(rta.capi.DEBUG_COOP_LEVEL, int(sys.argv[6]))
    if len(sys.argv) > 7:
        rta.capi.debug_set(rta.capi.DEBUG_NARROW_MAX, int(sys.argv[7]))
    if len(sys.argv) > 8:
        rta.capi.debug_set(rta.capi.DEBUG_NARROW_L2, int(sys.argv[8]))
    # any other control of csrc/rt_debug.h by name: WAVE_TIMELINE_KNOBS="COOP_MAX=1024,NARROW_MAX=32"
    for kv in filter(None, os.environ.get("WAVE_TIMELINE_KNOBS", "").split(",")):
        k, v = kv.split("=")
        rta.capi.debug_set(getattr(rta.capi, "DEBUG_" + k.strip().upper()), int(v))
    path = os.path.join(ROOT, "gpurun_out", "wave_trace.bin")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rta.capi.debug_set(rta.capi.DEBUG_ASYNC_ORDERS, 0)        # the dispatch orders at once, not from the background
    scene = rta.Scene.default(level)
    dev = scene.device(0)
    opts = (w, h, spp)
    regs_c = dev._regions([tuple(r) for r in rta.buckets(rta.RenderOptions(*opts))])
    out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(60):         # (the library times its candidate dispatch orders over the first launches of a tile list: let it settle)
        dev.render_frame_device(opts, regs_c, out.data_ptr(), stream)
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    rta.capi.wave_trace(path)
    dev.render_frame_device(opts, regs_c, out.data_ptr(), stream)
    torch.cuda.synchronize()
    rta.capi.wave_trace(None)
    rec = np.fromfile(path, dtype=np.uint32).reshape(-1, 8)          # rt_skip.hpp: start, end, HW_ID | XCC_ID << 16, descriptor, entry, ack, 0, 0
    ran = rec[:, 1] != 0
    r = rec[ran]
    t0 = r[:, 0].min()
    start = (r[:, 0] - t0).astype(np.int64) * 10          # ns
    end = (r[:, 1] - t0).astype(np.int64) * 10
    dur = end - start
    print("%dx%d spp %d L%d: %d waves ran (of %d slots), span %.1f us" % (w, h, spp, level, len(r), len(rec), end.max() / 1e3))
    print("wave duration us: median %.2f  p90 %.2f  p99 %.2f  max %.2f ; sum %.0f us = %.1f us x 1024 SIMDs" % (
        np.median(dur) / 1e3, np.percentile(dur, 90) / 1e3, np.percentile(dur, 99) / 1e3, dur.max() / 1e3, dur.sum() / 1e3, dur.sum() / 1e3 / 1024))
    print("last wave dispatched at %.1f us; first wave ends at %.1f us" % (start.max() / 1e3, end.min() / 1e3))
    coop = (r[:, 3] >> 31) != 0
    for name, sel in (("cooperative (rt_coop.hpp)", coop), ("skip-pointer walk", ~coop)):
        if sel.any():
            d = dur[sel]
            print("  %-26s %6d waves: median %.2f  p90 %.2f  p99 %.2f  max %.2f us; sum %.0f us; last end %.1f us" % (
                name, int(sel.sum()), np.median(d) / 1e3, np.percentile(d, 90) / 1e3, np.percentile(d, 99) / 1e3, d.max() / 1e3, d.sum() / 1e3, end[sel].max() / 1e3))
    if os.environ.get("WAVE_TIMELINE_JSON"):
        # what bench.py prints as roofline.longest_wave_us / span_us (profiles/roofline_waves.json, stamped with the kernel sources)
        import json
        from bench import kernel_src_sha, git_head
        jp = os.environ["WAVE_TIMELINE_JSON"]
        d = json.load(open(jp)) if os.path.exists(jp) else {}
        if d.get("kernel_src_sha") != kernel_src_sha():
            d = {}
        d.update({"kernel_src_sha": kernel_src_sha(), "git_head": git_head(), "tag": os.environ.get("WAVE_TIMELINE_TAG"),
                  "_source": "tools/wave_timeline.py: one traced launch (the -DRT_TEST_HOOKS build's trace flavour of the loops)"})
        if (w, h, spp, level) == (1920, 1080, 1, 8):
            d["k_render_skip_n1"] = {"span_us": round(end.max() / 1e3, 2), "longest_wave_us": round(dur.max() / 1e3, 2), "waves": int(len(r)),
                                     "median_wave_us": round(float(np.median(dur)) / 1e3, 2), "p99_wave_us": round(float(np.percentile(dur, 99)) / 1e3, 2),
                                     "waves_longer_than_0_9_of_the_span": int((dur > 0.9 * end.max()).sum()),
                                     "slot_time_over_8192_slots_us": round(dur.sum() / 1e3 / 8192, 2), "last_wave_start_us": round(start.max() / 1e3, 2)}
        json.dump(d, open(jp, "w"), indent=1, sort_keys=True)
    print("waves longer than 15 / 20 / 25 / 30 / 35 us: " + " / ".join(str(int((dur > t * 1000).sum())) for t in (15, 20, 25, 30, 35))
          + "; sum of wave time over 8,192 slots %.1f us" % (dur.sum() / 1e3 / 8192))
    # What a wave SLOT does between two waves (round 6).  A slot is (XCD = workgroup index % 8 -- the dispatcher deals workgroups round-robin
    # over the XCDs --, SE, CU, SIMD, wave id of HW_ID); per slot, consecutive occupants: prologue = start - entry (kernel arguments, descriptor),
    # ack = the pixel store's acknowledgement after `end` (s_endpgm waits for it), relaunch = next entry - this ack (the hardware's part).
    idx_all = np.nonzero(ran)[0]
    wg_i = idx_all // 4
    hw16 = (r[:, 2] & 0xFFFF).astype(np.int64)
    slot_id = ((((wg_i % 8) * 8 + ((hw16 >> 13) & 7)) * 16 + ((hw16 >> 8) & 15)) * 4 + ((hw16 >> 4) & 3)) * 16 + (hw16 & 15)
    entry = (r[:, 4].astype(np.int64) - int(t0)) * 10
    ack = (r[:, 5].astype(np.int64) - int(t0)) * 10
    o = np.lexsort((start, slot_id))
    same = slot_id[o][1:] == slot_id[o][:-1]
    relaunch = (entry[o][1:] - ack[o][:-1])[same]
    gap = (start[o][1:] - end[o][:-1])[same]
    pro = start - entry
    akl = ack - end
    med = lambda a: float(np.median(a)) / 1e3
    print("wave slots %d; between two waves of a slot (us, median / mean): end -> next start %.2f / %.2f = store ack %.2f / %.2f + relaunch (hardware) %.2f / %.2f + prologue %.2f / %.2f" % (
        len(np.unique(slot_id)), med(gap), gap.mean() / 1e3, med(akl), akl.mean() / 1e3, med(relaunch), relaunch.mean() / 1e3, med(pro), pro.mean() / 1e3))
    print("slot time between waves: %.0f us = %.1f %% of the waves' own %.0f us" % (gap.sum() / 1e3, 100.0 * gap.sum() / dur.sum(), dur.sum() / 1e3))
    order = np.argsort(-dur)[:16]
    print("16 longest waves: (start, end, dur us, dispatch index)")
    idx = np.nonzero(ran)[0]
    for k in order:
        print("   %.1f  %.1f  %.1f   #%d%s" % (start[k] / 1e3, end[k] / 1e3, dur[k] / 1e3, idx[k], " coop" if coop[k] else ""))
    # waves in flight over time
    T = int(end.max() // 1000) + 1
    print("time(us)  waves in flight   started so far")
    for t in range(0, T, max(1, T // 25)):
        ns = t * 1000
        print("  %4d     %6d           %6d" % (t, int(((start <= ns) & (end > ns)).sum()), int((start <= ns).sum())))
    hw = r[:, 2] & 0xFFFF
    xcc = r[:, 2] >> 16
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    se = (hw >> 13) & 7
    slot = ((xcc.astype(np.int64) * 8 + se) * 16 + cu) * 4 + simd
    busy = np.bincount(slot, weights=dur)
    busy = busy[busy > 0]
    print("SIMDs used %d; per-SIMD sum of wave time us: min %.1f median %.1f max %.1f" % (len(busy), busy.min() / 1e3, np.median(busy) / 1e3, busy.max() / 1e3))
    last_end = np.zeros(slot.max() + 1)
    np.maximum.at(last_end, slot, end)
    le = last_end[last_end > 0]
    print("per-SIMD time of last wave end us: min %.1f median %.1f max %.1f" % (le.min() / 1e3, np.median(le) / 1e3, le.max() / 1e3))


if __name__ == "__main__":
    main()
