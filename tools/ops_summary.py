"""Per-kind summary of the per-op HIP-event timings tools/profile_unet.py leaves under gpurun_out/.
Usage: python tools/ops_summary.py <label>=<ops.json> ... > profiles/rNN_unet_step_ops_<what>.txt"""
import json
import sys
from collections import defaultdict

KIND = {1: "gemm", 2: "groupnorm", 3: "layernorm", 4: "attention", 5: "softmax", 6: "to_cl", 7: "from_cl",
        8: "time_embed", 9: "copy2d", 10: "ddim_step", 11: "memset", 12: "lincomb", 13: "relpos_attn"}


def main(argv):
    print("per-op HIP-event timings (tools/profile_unet.py), one forward each; TF/s = algorithmic FLOP / event time")
    for arg in argv:
        label, path = arg.rsplit("=", 1)
        ops = json.load(open(path))
        tot = sum(o["ms"] for o in ops)
        fl = sum(o["flops"] for o in ops)
        print(f"# {label}: {len(ops)} ops, sum(op events) {tot:.2f} ms, {fl / 1e12:.2f} TFLOP -> "
              f"{fl / tot / 1e9:.1f} TF/s = {fl / tot / 1e9 / 2500:.3f} of the 2.5 PF/s dense fp16 peak")
        by = defaultdict(lambda: [0.0, 0, 0.0])
        for o in ops:
            k = KIND.get(o["kind"], str(o["kind"]))
            if k == "gemm":
                k += {0: "/plain", 1: "/conv3x3", 2: "/tconv", 3: "/conv_c8"}.get(o["meta"].get("gather", 0), "/conv(vae)")
            by[k][0] += o["ms"]
            by[k][1] += 1
            by[k][2] += o["flops"]
        for k, v in sorted(by.items(), key=lambda x: -x[1][0]):
            print(f"  {k:<14} {v[0]:8.3f} ms {100 * v[0] / tot:5.1f}% {v[1]:4d} ops {v[2] / max(v[0], 1e-9) / 1e9:8.1f} TF/s")


if __name__ == "__main__":
    main(sys.argv[1:])
