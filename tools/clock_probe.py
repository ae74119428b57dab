"""Sample shader clock / socket power while the bench forward loop runs (diagnostic)."""
import os, sys, time, threading, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
samples = []; stop = False
def sampler():
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[0]
        while not stop:
            rec = {"t": time.time()}
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                for k in ("current_gfxclk", "average_gfxclk_frequency", "current_socket_power", "average_socket_power",
                          "temperature_hotspot", "current_uclk", "throttle_status", "indep_throttle_status"):
                    if k in m: rec[k] = m[k]
                if "current_gfxclks" in m: rec["gfxclks"] = m["current_gfxclks"][:8]
            except Exception as e:
                rec["err"] = repr(e)
            samples.append(rec); time.sleep(0.02)
    except Exception as e:
        samples.append({"fatal": repr(e)})
th = threading.Thread(target=sampler, daemon=True); th.start()
sys.argv = ["bench.py", "--steps", os.environ.get("PROBE_STEPS", "300"), "--warmup", "3", "--no-cpu-baseline"] + sys.argv[1:]
try:
    import bench
    bench.main()
finally:
    stop = True; th.join()
t0 = samples[0].get("t", 0) if samples else 0
for r in samples[:: max(1, len(samples) // 60)]:
    if "t" in r: r["t"] = round(r["t"] - t0, 2)
    print(json.dumps(r))
