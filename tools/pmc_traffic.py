"""Per-kernel HBM traffic per launch from the two PMC passes of tools/profile_round.sh
(gpurun_out/pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE.txt) -> JSON {kernel name: {fetch_kb, write_kb, launches}}.
Usage: python tools/pmc_traffic.py gpurun_out profiles/r01_pmc_traffic.json"""
import json
import re
import sys


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"\s*(\d+)\s+([0-9.]+)\s+(.*\S)\s*$", line)
        if m:
            out[m.group(3)] = (int(m.group(1)), float(m.group(2)))
    return out


def main(src, dst):
    f, w = parse(f"{src}/pmc_FETCH_SIZE.txt"), parse(f"{src}/pmc_WRITE_SIZE.txt")
    res = {}
    for k in sorted(set(f) | set(w)):
        res[k] = {"fetch_kb": f.get(k, (0, 0.0))[1], "write_kb": w.get(k, (0, 0.0))[1], "launches": f.get(k, w.get(k))[0]}
    json.dump({"unit": "KB per launch, raw rocprofv3 FETCH_SIZE / WRITE_SIZE (separate --pmc passes)",
               "note": "gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                       "(MI355X_MICROARCH.md, HBM); uncalibrated for other access widths",
               "kernels": res}, open(dst, "w"), indent=1)
    print("wrote", dst, len(res), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
