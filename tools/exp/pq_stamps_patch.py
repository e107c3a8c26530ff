#!/usr/bin/env python3
"""Diagnostic build of gemm_pq_kernel with in-kernel s_memtime stamps (never shipped): writes a patched copy of gemm_big.hip.
   python tools/exp/pq_stamps_patch.py OUT.hip ; hipcc ... -c OUT.hip ; link with the other objects into _exp_lib/ ;
   read the stamps with mh_exp_pq_stamps() (tools/exp/pq_stamps_read.py)."""
import sys, re, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "mirror_amd/csrc/gemm_big.hip")).read()
def sub(old, new, count=1):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new, count)
NS = 640
# stamp storage + macro, in front of the pq epilogues
sub("constexpr int PQ_SLOT = 68 * 1024;", f"""constexpr int PQ_NS = {NS};
__device__ unsigned pq_stamps_dev[2 * 8 * PQ_NS];
#define PQ_STAMP() do {{ unsigned long long t_; asm volatile("s_memtime %0\\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \\
    if (stamp_on && stamp_cnt < PQ_NS) {{ if ((threadIdx.x & 63) == 0) stamp_base[stamp_cnt] = (unsigned)t_; stamp_cnt++; }} }} while (0)
constexpr int PQ_SLOT = 68 * 1024;""")
# epilogue bf16: stamps around its phases
sub("                                                 AfterRead&& after_read) {\n    constexpr int PITCH = BIG + 4, HALF = BIG / 2;",
    "                                                 AfterRead&& after_read, bool stamp_on, unsigned* stamp_base, int& stamp_cnt) {\n    constexpr int PITCH = BIG + 4, HALF = BIG / 2;")
sub("    // physical rows of this tile (row windows, GemmArgs.c_rpb)", "    PQ_STAMP();      // packed\n    // physical rows of this tile (row windows, GemmArgs.c_rpb)")
sub("""        if (half == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        constexpr int CPR = BIG / 8;""", """        PQ_STAMP();      // written (or idle hook done)
        if (half == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PQ_STAMP();      // waited
        __syncthreads();
        PQ_STAMP();      // past barrier
        constexpr int CPR = BIG / 8;""")
sub("""        __syncthreads();
        if (half == 1 && MODE == 0) after_read();""", """        PQ_STAMP();      // staging read
        __syncthreads();
        PQ_STAMP();      // past barrier
        if (half == 1 && MODE == 0) after_read();""")
sub("""        if (half == 1 && MODE != 0) after_read();
    }
    if constexpr (EPI == MH_EPI_SQERR) {
        float* red""", """        if (half == 1 && MODE != 0) after_read();
        PQ_STAMP();      // stores issued
    }
    if constexpr (EPI == MH_EPI_SQERR) {
        float* red""")
src = re.sub(r"(pq_epilogue_bf16<0, (?:EPI|0), VAR>\([^;]*?after_read)\)", r"\1, stamp_on, stamp_base, stamp_cnt)", src)
# kernel: state
sub("    int v = blockIdx.x;\n    if (v >= units) return;", """    const bool stamp_on = blockIdx.x == 0 || blockIdx.x == 129;
    unsigned* stamp_base = reinterpret_cast<unsigned*>(smem + PQ_LDS) + wave * PQ_NS;
    int stamp_cnt = 0;
    int v = blockIdx.x;
    if (v >= units) return;""")
sub("                const char* asub = cur + kh * P2_SUB;", "                PQ_STAMP();\n                const char* asub = cur + kh * P2_SUB;")
sub("""                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);""", """                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                PQ_STAMP();
                __builtin_amdgcn_s_barrier();
                PQ_STAMP();
                __builtin_amdgcn_sched_barrier(0);""")
sub("""                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();          // both wave rows level again""", """                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                PQ_STAMP();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PQ_STAMP();
        if (wm == 0) __builtin_amdgcn_s_barrier();          // both wave rows level again""")
# dump at kernel end: find the end of the unit loop = "sa = sa_n; sb = sb_n;\n    }\n}"
sub("        lnext = false;\n    }\n}", """        lnext = false;
        PQ_STAMP();
    }
    if (stamp_on) {
        __syncthreads();
        const unsigned* st = reinterpret_cast<const unsigned*>(smem + PQ_LDS);
        for (int i = tid; i < 8 * PQ_NS; i += NTB) pq_stamps_dev[(blockIdx.x == 0 ? 0 : 8 * PQ_NS) + i] = st[i];
    }
}
extern "C" int mh_exp_pq_stamps(unsigned* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pq_stamps_dev), sizeof(unsigned) * 2 * 8 * PQ_NS); }""")
# LDS size of the launch
sub("dim3(pq_grid(units_)), dim3(NTB), PQ_LDS, s, a, (int)units_,", f"dim3(pq_grid(units_)), dim3(NTB), PQ_LDS + 8 * PQ_NS * 4, s, a, (int)units_,")
sub("hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);", "hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);")
# timing experiments (results are garbage): NOREQ = no stage requests behind the prologue, NOREAD = one fragment read per segment
if os.environ.get("PQ_EXP") == "NOREQ":
    sub("const bool do_issue = !(last && kh == 1) && (!lnext || has_next);", "const bool do_issue = false;")
if os.environ.get("PQ_EXP") == "NOREAD":
    sub("if constexpr (AKC) af[ks][i] = p2_frag_kc(asub, offa2, i, ks);", "if constexpr (AKC) af[ks][i] = p2_frag_kc(asub, offa2, 0, 0);")
    sub("if constexpr (BKC) bfr[ks][j] = p2_frag_kc(bsub, offb2, j, ks);", "if constexpr (BKC) bfr[ks][j] = p2_frag_kc(bsub, offb2, 0, 0);")
open(sys.argv[1], "w").write(src)
