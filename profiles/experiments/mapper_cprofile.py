"""Host-side profile of the mapper loop (cProfile, sorted by own time): python profiles/experiments/mapper_cprofile.py"""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("ml", "examples/mapper_loop.py"); ml = importlib.util.module_from_spec(spec); spec.loader.exec_module(ml)
pr = cProfile.Profile()
pr.enable(); ml.main(); pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45); print(s.getvalue()[:9000])
