#!/bin/bash
# Instrumented diagnostic copy of the library (scripts/bin/libpyrad_hip_phase.so): every wave of the far-field
# accumulate kernel stamps s_memrealtime at entry, after the edge lines, after the series phase, after the near lines
# and at exit.  The real library never executes a stamp.  Use with scripts/phase_times.py.
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
head = "template <int R, int LS, int NT = 0>"
assert head in s
s = s.replace(head, "__device__ unsigned long long g_dbg[4 * 65536 * 6];\n\n" + head, 1)
def rep(a, b):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b, 1)
rep("    constexpr bool FF = NT > 0;\n", "    constexpr bool FF = NT > 0;\n    unsigned long long t_ph[5] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0, 0};\n")
rep("        if (any_far) {\n            // the running fraction of the edge lines", "        t_ph[1] = __builtin_amdgcn_s_memrealtime();\n        if (any_far) {\n            // the running fraction of the edge lines")
rep("        // the direct classes: left-edge, near and right-edge lines.", "        t_ph[2] = __builtin_amdgcn_s_memrealtime();\n        // the direct classes: left-edge, near and right-edge lines.")
rep("    // Results leave through LDS so that every store instruction writes 512 contiguous bytes\n    // (a lane owns R CONSECUTIVE points;", "    t_ph[3] = __builtin_amdgcn_s_memrealtime();\n    // Results leave through LDS so that every store instruction writes 512 contiguous bytes\n    // (a lane owns R CONSECUTIVE points;")
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
        const size_t k = ((size_t)blockIdx.x * 4 + (wave & 3)) * 6;
        g_dbg[k] = t_ph[0]; g_dbg[k + 1] = t_ph[1]; g_dbg[k + 2] = t_ph[2]; g_dbg[k + 3] = t_ph[3];
        g_dbg[k + 4] = __builtin_amdgcn_s_memrealtime(); g_dbg[k + 5] = ((unsigned long long)xcc << 32) | hwid;
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
cp $T/pyrad_amd/lib/libpyrad_hip.so $ROOT/scripts/bin/libpyrad_hip_phase.so
rm -rf $T
echo built scripts/bin/libpyrad_hip_phase.so
