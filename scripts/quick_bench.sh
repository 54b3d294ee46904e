# one line per bench invocation: step and kernel times (run on the GPU box); usage: quick_bench.sh "label" <bench args...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
label=$1; shift
python3 $R/bench.py --steps 30 --warmup 3 --blocks 3 --no-cpu-baseline --no-api-path --no-direct-pass --legs none "$@" 2>/dev/null |
  python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', 'step %.4f ms' % d['ms_per_step'], 'blocks', d['ms_per_step_blocks']['blocks'], {k: round(v, 4) for k, v in d['kernel_ms_per_step'].items() if k != 'source'})"
