"""Index construction time with the suffix arrays on the device vs on the host. usage: index_build_time.py genome_bp [host]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import floxer_amd as F
from floxer_amd import simulate as S
G = int(sys.argv[1])
genome = S.make_genome(G, 1, seed=S.DEFAULT_SEED)
t = time.time(); d = F.fmindex(genome, device=0); td = time.time() - t
print(f"{G} bp: device-built index {td:.1f} s", flush=True)
if len(sys.argv) > 2:
    t = time.time(); h = F.fmindex(genome); th = time.time() - t
    print(f"{G} bp: host-built index {th:.1f} s", flush=True)
    d.save("/tmp/d.idx"); h.save("/tmp/h.idx")
    import filecmp
    print("files equal:", filecmp.cmp("/tmp/d.idx", "/tmp/h.idx", shallow=False))
