"""HBM traffic per decode step from rocprofv3 PMC passes (one counter per pass).

usage: python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv>

Per MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read (16 B/lane), so it is doubled;
WRITE_SIZE is exact.  A decode step = the dispatches from one sampler_kernel to the next.
"""
import csv
import json
import sys


def per_step(path, counter):
    rows = list(csv.DictReader(open(path)))
    rows = [r for r in rows if r.get("Counter_Name") == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    steps, cur = [], None
    for r in rows:
        name = r["Kernel_Name"]
        if "sampler_kernel" in name:
            if cur is not None:
                steps.append(cur)
            cur = {}
        if cur is None:
            continue            # prefill dispatches before the first sampler launch
        if "attn_split" in name:
            key = "attn_split"
        elif "sampler" in name:
            key = "sampler"
        elif "gemv_kernel<" in name:
            key = "gemv<" + name.split("gemv_kernel<")[1].split(">(")[0].replace("t3k::(anonymous namespace)::", "") + ">"
        else:
            key = name[:40]
        cur[key] = cur.get(key, 0.0) + float(r["Counter_Value"])
    return steps


def main():
    fetch = per_step(sys.argv[1], "FETCH_SIZE")
    write = per_step(sys.argv[2], "WRITE_SIZE")
    n = min(len(fetch), len(write))
    fetch, write = fetch[-min(n, 10):], write[-min(n, 10):]
    f_kib = sum(sum(s.values()) for s in fetch) / len(fetch)
    w_kib = sum(sum(s.values()) for s in write) / len(write)
    out = {
        "steps_averaged": len(fetch),
        "FETCH_SIZE_KiB_per_step_raw": round(f_kib, 1),
        "WRITE_SIZE_KiB_per_step": round(w_kib, 1),
        "hbm_read_bytes_per_step_corrected_x2": int(2 * f_kib * 1024),
        "hbm_write_bytes_per_step": int(w_kib * 1024),
        "hbm_bytes_per_step": int((2 * f_kib + w_kib) * 1024),
        "per_kernel_fetch_KiB_raw": {k: round(sum(s.get(k, 0) for s in fetch) / len(fetch), 1) for k in fetch[-1]},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
