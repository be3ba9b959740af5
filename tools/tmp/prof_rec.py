import os, sys, types, cProfile, pstats, torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle"); sys.path.insert(0, ROOT + "/overlay")
from robir_amd import renderer
from test_deferred_gpu import _view, split_input, plot_loop
from test_runner_hooks_gpu import make_pbr_runner_hook
dev = torch.device("cuda:0")
with torch.no_grad():
    model = renderer.build_synthetic_model(dev)
    model.deferred_chunks = 1024
    mi, total = _view(dev, 800, 800)
    for rep in range(2):
        split = split_input(mi, total)
        pr = cProfile.Profile()
        pr.enable()
        merged = plot_loop(model, split, total)
        pr.disable()
        merged["pred_rgb"].cpu()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
