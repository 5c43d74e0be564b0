cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; }
for A in "--config c5 --steps 300 --warmup 30" "--config big --steps 300 --warmup 30" "--envs-per-gpu 131072 --steps 500 --warmup 50"; do
  echo -n "[$A] placed: "; for i in 1 2 3 4 5 6; do python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line; done; echo
  echo -n "[$A] first allocation: "; for i in 1 2 3 4 5 6; do RISVEC_NO_PLACEMENT=1 python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line; done; echo
done
python -c "
import torch, sys
sys.path.insert(0, '.')
from bench import build_env
e = build_env(32768, 16, 256, torch.device('cuda:0'), 0, 0); print(e.placement)
e = build_env(262144, 8, 64, torch.device('cuda:0'), 0, 0); print(e.placement)
e = build_env(100000, 8, 100, torch.device('cuda:0'), 0, 0); print(e.placement)
"
