#!/usr/bin/env python3
"""Would the target-pass ViT forward hide under the generator forward?  (round 6 experiment, tools only)

Today a step starts with netG(A crop) || netG(B crop) on its two streams -- two chains of ~55 launches of 5-10 us each, the chip mostly
idle -- and only then runs the two ViT forwards (targets || generated) side by side.  The target passes do not depend on the generator.
If ONE generator call covered both crops (batch of two independent images: the launch count of one chain), the other stream could start
the target forward at t = 0.  Whether that pays depends on what the latency-bound generator kernels lose when GEMMs share the chip.

Measured here, each arm captured into ONE hipGraph (two branches where there are two) and replayed:
    a  netG(1 image)                         alone
    b  netG(1) || netG(1)                    today's opening
    c  netG(2 images, one chain)             alone
    d  ViT forward, 2 passes, no grad        alone            (the target passes)
    e  ViT forward, 2 passes || ViT forward, 2 passes          today's second phase
    f  netG(2) || ViT forward 2 passes                         the proposed opening
    g  today's forward:    [netG(1) || netG(1)]  then  [ViT 2 || ViT 2]
    h  proposed forward:   [netG(2) -> ViT 2 (generated)]  ||  [ViT 2 (targets)]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from splice_amd import synth
from splice_amd.generator import GeneratorEngine, GeneratorPlan
from splice_amd.vit import VitContext, VitEngine

DEV = "cuda"
S = int(os.environ.get("SIZE", "224"))


def timed_graph(build, reps=30):
    s0 = torch.cuda.Stream()
    with torch.cuda.stream(s0):
        build(s0)          # warm-up, eager
        s0.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s0):
            build(s0)
        for _ in range(5):
            g.replay()
        s0.synchronize()
        best = 1e30
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s0)
            for _ in range(reps):
                g.replay()
            b.record(s0)
            s0.synchronize()
            best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


def main():
    vit = VitEngine("dino_vitb8", device=DEV).load_state_dict(synth.vit_params(1234, "dino_vitb8", img_size=S))
    gen = GeneratorEngine(device=DEV)
    params = gen.flatten(synth.generator_params(1235, 0.02))
    p1a, p1b = GeneratorPlan(gen, 1, S, S, True), GeneratorPlan(gen, 1, S, S, True)
    p2 = GeneratorPlan(gen, 2, S, S, True)
    cA, cB = VitContext(vit, 2, S, S, True), VitContext(vit, 2, S, S, False)
    x1 = torch.rand(1, 3, S, S, device=DEV)
    x2 = torch.rand(2, 3, S, S, device=DEV)
    im2 = torch.rand(2, 3, S, S, device=DEV)
    side = torch.cuda.Stream()

    def fork(s0):
        side.wait_stream(s0)

    def join(s0):
        s0.wait_stream(side)

    arms = {}
    arms["a netG(1) alone"] = lambda s0: p1a.forward(params, x1)

    def b(s0):
        fork(s0)
        with torch.cuda.stream(side):
            p1b.forward(params, x1)
        p1a.forward(params, x1)
        join(s0)
    arms["b netG(1) || netG(1)"] = b
    arms["c netG(2) one chain"] = lambda s0: p2.forward(params, x2)
    arms["d ViT fwd 2 passes alone (no grad)"] = lambda s0: cB.forward(im2, True)

    def e(s0):
        fork(s0)
        with torch.cuda.stream(side):
            cB.forward(im2, True)
        cA.forward(im2, True)
        join(s0)
    arms["e ViT 2 || ViT 2"] = e

    def f(s0):
        fork(s0)
        with torch.cuda.stream(side):
            cB.forward(im2, True)
        p2.forward(params, x2)
        join(s0)
    arms["f netG(2) || ViT 2"] = f

    def g(s0):
        b(s0)
        e(s0)
    arms["g today:    [netG(1) || netG(1)] then [ViT 2 || ViT 2]"] = g

    def h(s0):
        fork(s0)
        with torch.cuda.stream(side):
            cB.forward(im2, True)
        y = p2.forward(params, x2)
        cA.forward(im2, True)
        join(s0)
    arms["h proposed: [netG(2) -> ViT 2] || [ViT 2]"] = h
    for name, fn in arms.items():
        print(f"{name:62s} {timed_graph(fn):8.1f} us", flush=True)


if __name__ == "__main__":
    main()
