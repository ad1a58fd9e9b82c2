"""profiles/rNN_shards.txt from the per-rank bench lines of a session (shard_all.json, shard_G_R.json written by
`bench.py --shard R/G --force-dist`):   python tools/shard_table.py gpurun_out/r4final > profiles/r04_shards.txt"""
import json
import os
import sys


def line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main(d):
    whole = line(os.path.join(d, "shard_all.json"))["ms_per_step"]
    print("# Every rank's share of BASELINE config 2 (N = 2^20 fp64 Morlet, 256 scales) measured on ONE GPU: python bench.py --steps 20")
    print("# --warmup 3 --shard R/G --force-dist (the rows rank R of G owns under the cost-balanced contiguous partition, with the 1-rank")
    print("# RCCL broadcast of the signal per step), one box.  No multi-GPU node was available: the speed-up column is what G such")
    print("# ranks would deliver if the xGMI broadcast hides behind the previous step -- a projection from single-GPU evidence,")
    print("# unmeasured on hardware.")
    print(f"all 256 rows on this box: {whole:.4f} ms per step\n")
    for G in (2, 4, 8):
        ranks = []
        for r in range(G):
            p = os.path.join(d, f"shard_{G}_{r}.json")
            if not os.path.exists(p):
                break
            x = line(p)
            ranks.append((x["ms_per_step"], {k: v for k, v in x["roofline"]["row_split"].items() if v}))
        if len(ranks) != G:
            continue
        slow, total = max(t for t, _ in ranks), sum(t for t, _ in ranks)
        print(f"G = {G}: slowest rank {slow:.3f} ms -> {whole / slow:.2f}x   (sum of the ranks {total:.3f} ms: "
              f"{(total - whole) / G * 1e3:.0f} us of fixed cost per rank; perfectly balanced {whole / (total / G):.2f}x)")
        for r, (t, split) in enumerate(ranks):
            print(f"    rank {r}: {t:.3f} ms  {split}")


if __name__ == "__main__":
    main(sys.argv[1])
