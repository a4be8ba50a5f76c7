import cProfile, pstats, sys, io, torch
sys.path.insert(0, '.')
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-kernel-events', '--steps', '15', '--warmup', '3']
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats('tottime')
st.print_stats(45)
print(s.getvalue()[:9000])
s2 = io.StringIO()
pstats.Stats(pr, stream=s2).sort_stats('cumulative').print_stats('ops.py|parallel.py|train.py|rnn_|syntax_vae', 40)
print(s2.getvalue()[:9000])
