#!/usr/bin/env python3
"""Per-launch breakdown of ONE steady-state tick from a rocprofv3 kernel trace, annotated with the
fused plan's layer shapes (FLOPs, minimal bytes) -> achieved TFLOP/s and GB/s per layer."""
import csv, glob, sys
sys.path.insert(0, ".")

def plan_layers(scale="s", B=32):
    import torch
    from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net, ConvBnAct, C2f, SPPF
    net = build_detector_net(scale).fuse()
    L = []
    def conv(mod, h, w, name):
        c = mod.conv if isinstance(mod, ConvBnAct) else mod
        k, s = c.kernel_size[0], c.stride[0]
        ho, wo = (h - 1)//s + 1 if k == 3 else h//s, (w - 1)//s + 1 if k == 3 else w//s
        M = B*ho*wo
        fl = 2*M*c.out_channels*c.in_channels*k*k
        by = 2*(B*h*w*c.in_channels + M*c.out_channels)
        L.append((name, f"{c.in_channels}->{c.out_channels} k{k}s{s} {h}x{w}", fl, by))
    def c2f(mod, h, w, name):
        conv(mod.cv1, h, w, name+".cv1")
        for i, b in enumerate(mod.m):
            conv(b.cv1, h, w, f"{name}.m{i}.cv1"); conv(b.cv2, h, w, f"{name}.m{i}.cv2")
        conv(mod.cv2, h, w, name+".cv2")
    L.append(("stem", "3->%d k3s2 640x640" % net.b0.conv.out_channels, 2*B*320*320*net.b0.conv.out_channels*27, 2*B*(3*640*640+320*320*net.b0.conv.out_channels)))
    conv(net.b1, 320, 320, "b1"); c2f(net.b2, 160, 160, "b2"); conv(net.b3, 160, 160, "b3"); c2f(net.b4, 80, 80, "b4")
    conv(net.b5, 80, 80, "b5"); c2f(net.b6, 40, 40, "b6"); conv(net.b7, 40, 40, "b7"); c2f(net.b8, 20, 20, "b8")
    conv(net.b9.cv1, 20, 20, "sppf.cv1"); [L.append((f"sppf.pool{i}", "", 0, 0)) for i in range(3)]; conv(net.b9.cv2, 20, 20, "sppf.cv2")
    L.append(("up5", "", 0, 0)); c2f(net.h12, 40, 40, "h12"); L.append(("up4", "", 0, 0)); c2f(net.h15, 80, 80, "h15")
    conv(net.h16, 80, 80, "h16"); c2f(net.h18, 40, 40, "h18"); conv(net.h19, 40, 40, "h19"); c2f(net.h21, 20, 20, "h21")
    for lvl, hw in enumerate((80, 40, 20)):
        for br, seq in (("box", net.detect.box[lvl]), ("cls", net.detect.cls[lvl])):
            for j, m in enumerate(seq):
                conv(m, hw, hw, f"det{lvl}.{br}{j}")
        L.append((f"head{lvl}", "", 0, 0))
    return L

def main(src, scale="s"):
    f = sorted(glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k1_ratio" in r["Kernel_Name"]]
    a, b = idx[-3], idx[-2]
    tick = rows[a:b]
    layers = plan_layers(scale)
    det = [r for r in tick if any(k in r["Kernel_Name"] for k in ("k_conv_mfma", "k_conv_res", "k_stem", "k_maxpool5", "k_upsample2", "k_head"))]
    print(f"tick launches: {len(tick)}, detector launches: {len(det)} (plan {len(layers)})")
    tot = 0
    for r, (name, desc, fl, by) in zip(det, layers):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3
        tot += us
        kn = r["Kernel_Name"].split("::")[-1][:26]
        print(f"{name:14s} {desc:28s} {kn:26s} {us:8.1f} us  {fl/us/1e6 if us else 0:7.1f} TF/s  {by/us/1e3 if us else 0:7.0f} GB/s(min-bytes)")
    print("detector kernel time %.1f us; tick span %.1f us" % (tot, (int(tick[-1]["End_Timestamp"]) - int(tick[0]["Start_Timestamp"]))/1e3))

if __name__ == "__main__":
    main(*sys.argv[1:])
