#!/bin/bash
# Build the instrumented diagnostic copy of the library (scripts/bin/libpyrad_hip_dbg.so): every wave of
# the LS accumulate kernel stamps s_memrealtime at entry and exit plus its XCC / HW_ID.  The real
# library never executes a stamp.  Use with scripts/wave_times.py:
#   PYRAD_HIP_LIB=$PWD/scripts/bin/libpyrad_hip_dbg.so python scripts/wave_times.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/pyrad_amd/csrc $T/include $ROOT/scripts/bin
cp $ROOT/pyrad_amd/csrc/* $T/pyrad_amd/csrc/
cp $ROOT/include/pyrad_hip.h $T/include/
python3 - "$T" <<'PY'
import sys
T = sys.argv[1]
p = T + '/pyrad_amd/csrc/lbl_kernels.hip'
s = open(p).read()
head = "#ifndef LBL_FF_MIN_WAVES"
if head not in s:
    head = "template <int R, int LS, bool FF = false>\n__global__"
assert head in s
s = s.replace(head, "__device__ unsigned long long g_dbg[4 * 65536 * 3];\n\n" + head, 1)
s = s.replace("    int job = blockIdx.y, tile;\n", "    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();\n    int job = blockIdx.y, tile;\n", 1)
marker = """                    fused_finish(J.fuse, wlo + o, kk);
                }
            }
        }
    }
}
"""
assert marker in s
s = s.replace(marker, marker[:-2] + '''    if (lane == 0 && blockIdx.x < 65536) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const size_t k = ((size_t)blockIdx.x * 4 + (wave & 3)) * 3;
        g_dbg[k] = t_start; g_dbg[k + 1] = __builtin_amdgcn_s_memrealtime(); g_dbg[k + 2] = ((unsigned long long)xcc << 32) | hwid;
    }
}
''', 1)
s += "\nnamespace lbl { void* dbg_symbol() { void* p = nullptr; (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_dbg)); return p; } }\n"
open(p, 'w').write(s)
p = T + '/pyrad_amd/csrc/lbl_api.hip'
s = open(p).read()
s += '''
namespace lbl { void* dbg_symbol(); }
extern "C" int lbl_debug_times(lbl_ctx* ctx, unsigned long long* out, int n) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpy(out, lbl::dbg_symbol(), (size_t)n * 8, hipMemcpyDeviceToHost);
    return n;
}
'''
open(p, 'w').write(s)
PY
make -C $T/pyrad_amd/csrc -j4 > $T/build.log 2>&1 || { grep -E "error" $T/build.log; exit 1; }
cp $T/pyrad_amd/lib/libpyrad_hip.so $ROOT/scripts/bin/libpyrad_hip_dbg.so
rm -rf $T
echo built scripts/bin/libpyrad_hip_dbg.so
