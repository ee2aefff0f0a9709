"""Write profiles/rNN_pmc_traffic.json (the `roofline.traffic` figure of bench.py) from two rocprofv3 PMC passes.

On a GPU box (counters in passes of their own, kernel-trace only -- see the guide's HBM / rocprofv3 section):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-graph
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d $R/gpurun_out/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-graph
    python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/r02_pmc_traffic.json
The file is stamped with the hash of the kernel sources it was measured on; bench.py refuses a file whose stamp differs."""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_stamp, workload constants)


def counter_avg(path, counter, name_filter):
    dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    assert dbs, "no rocpd database under " + path
    tot, n = 0.0, 0
    for dbp in dbs:
        cur = sqlite3.connect(dbp).cursor()
        # one row per (dispatch, counter): sum over the dispatch's instances first, then average over dispatches
        rows = cur.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? group by dispatch_id, kernel_name",
                           (counter,)).fetchall()
        for k, _, v in rows:
            if name_filter in k:
                tot += v
                n += 1
    return tot / max(n, 1), n


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    # optional: batch grid sample_steps of the profiled bench.py command (default: the headline workload)
    batch, grid, sample_steps = (int(v) for v in sys.argv[4:7]) if len(sys.argv) >= 7 else (1, 32, 8)
    # optional: the model name and the remaining bench.py flags of the profiled command (the 1B shares: "1b" "--s-byt5 256 --clip-image 1 [--inpaint]")
    model = sys.argv[7] if len(sys.argv) >= 8 else "570m"
    more = sys.argv[8] if len(sys.argv) >= 9 else ""
    extra = "" if (model, batch, grid, sample_steps) == ("570m", 1, 32, 8) else " --model %s --batch %d --grid %d --sample-steps %d %s" % (model, batch, grid, sample_steps, more)
    f_kb, n_f = counter_avg(fetch_dir, "FETCH_SIZE", "gemm_nt_kernel")
    w_kb, n_w = counter_avg(write_dir, "WRITE_SIZE", "gemm_nt_kernel")
    hbm = (2.0 * f_kb + w_kb) * 1024.0
    j = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and a separate pass --pmc WRITE_SIZE) -- python bench.py%s --steps N --warmup 1 --no-cpu-baseline --no-extra --no-graph" % extra,
         "workload": {"model": model, "batch_per_gpu": batch, "grid": grid, "sample_steps": sample_steps},
         "source_stamp": bench.source_stamp(), "stamped_sources": bench.TRAFFIC_SOURCES,
         "kernel": "gemm_nt_kernel (all instantiations)", "launches_profiled": n_f,
         "fetch_size_kb_avg_per_launch": round(f_kb, 2), "write_size_kb_avg_per_launch": round(w_kb, 2),
         "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is uncalibrated and taken as is; both are KiB",
         "hbm_bytes_per_launch": round(hbm)}
    assert n_f == n_w or abs(n_f - n_w) < 0.01 * n_f, (n_f, n_w)
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j, indent=1))


if __name__ == "__main__":
    main()
