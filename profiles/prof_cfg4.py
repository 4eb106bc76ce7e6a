import sys, time, cProfile, pstats, numpy as np
sys.path.insert(0, '.')
from bench import config4_structures, load_weights
from pesto_amd import Model, sharding
from pesto_amd.config import CONFIGS
cfg = CONFIGS["i_v4_1"]
m = Model(cfg, validate=False).to("cuda:0"); m.load_state_dict(load_weights(cfg)[0])
structs, sizes, _ = config4_structures(64, m)
sharding.forward_sharded(m, structs, 5)
t = time.perf_counter(); sharding.forward_sharded(m, structs, 5); print("pass", time.perf_counter() - t)
pr = cProfile.Profile(); pr.enable(); sharding.forward_sharded(m, structs, 5); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
