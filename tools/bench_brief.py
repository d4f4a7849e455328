import json, sys
for p in sys.argv[1:]:
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(p, "Mpx/s %.0f ms/step %.4f light_us %.1f frac %.3f host_ms %.3f" % (d["value"], d["ms_per_step"], r.get("avg_launch_us", 0), r.get("frac", 0), d.get("host_busy_ms_per_step", 0)))
    except Exception as e:  # noqa
        print(p, "unreadable:", e)
