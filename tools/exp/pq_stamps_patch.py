#!/usr/bin/env python3
"""Diagnostic build of gemm_pq_kernel with in-kernel s_memtime stamps (never shipped): writes a patched copy of gemm_big.hip.
   python tools/exp/pq_stamps_patch.py OUT.hip
   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Imirror_amd/csrc -Iinclude -c OUT.hip -o OUT.o
   hipcc --offload-arch=gfx950 -shared -fPIC -o _exp_lib/libmirror_stamps.so <mirror_amd/csrc/build/*.o without gemm_big.o> OUT.o
   MIRROR_HIP_LIB=$PWD/_exp_lib/libmirror_stamps.so python tools/exp/pq_stamps_read.py [N] [K]
   Every wave of workgroups 0 and 129 stores the shader clock into LDS at each phase boundary of the bf16 path (K loop segment: start,
   waited, past barrier 1, MFMAs issued; epilogue: packed, then per half written / past barrier / staging read / past barrier / stores
   issued) and the workgroup dumps them at kernel end (mh_exp_pq_stamps)."""
import sys, re, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "mirror_amd/csrc/gemm_big.hip")).read()
def sub(old, new, count=1):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new, count)
NS = 640
sub("constexpr int PQ_SLOT = 68 * 1024;", f"""constexpr int PQ_NS = {NS};
__device__ unsigned pq_stamps_dev[2 * 8 * PQ_NS];
#define PQ_STAMP() do {{ unsigned long long t_; asm volatile("s_memtime %0\\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \\
    if (stamp_on && stamp_cnt < PQ_NS) {{ if ((threadIdx.x & 63) == 0) stamp_base[stamp_cnt] = (unsigned)t_; stamp_cnt++; }} }} while (0)
constexpr int PQ_SLOT = 68 * 1024;""")
# ---- bf16 epilogue
sub("int tile_row0, int tile_col0, int wm, int wn, int lane, int tid, bool has_bias, float alpha) {\n    constexpr int PITCH = BIG + 4, HALF = BIG / 2;",
    "int tile_row0, int tile_col0, int wm, int wn, int lane, int tid, bool has_bias, float alpha, bool stamp_on, unsigned* stamp_base, int& stamp_cnt) {\n    constexpr int PITCH = BIG + 4, HALF = BIG / 2;")
sub("    // physical rows of this tile (row windows, GemmArgs.c_rpb)", "    PQ_STAMP();      // packed\n    // physical rows of this tile (row windows, GemmArgs.c_rpb)")
sub("""        __syncthreads();
        constexpr int CPR = BIG / 8;                 // 16-byte chunks per tile row""", """        PQ_STAMP();      // written
        __syncthreads();
        PQ_STAMP();      // past barrier
        constexpr int CPR = BIG / 8;                 // 16-byte chunks per tile row""")
sub("""            o[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        __syncthreads();""", """            o[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        PQ_STAMP();      // staging read
        __syncthreads();
        PQ_STAMP();      // past barrier""")
sub("""                    sq_cnt += 8.f;
                }
            }
        }
    }""", """                    sq_cnt += 8.f;
                }
            }
        }
        PQ_STAMP();      // stores issued
    }""")
src = re.sub(r"(pq_epilogue_bf16<0, (?:EPI|0), VAR>\([^;]*?g\.alpha)\)", r"\1, stamp_on, stamp_base, stamp_cnt)", src)
# ---- kernel
sub("    int v = blockIdx.x;\n    if (v >= units) return;", """    const bool stamp_on = blockIdx.x == 0 || blockIdx.x == 129;
    unsigned* stamp_base = reinterpret_cast<unsigned*>(smem + PQ_LDS) + wave * PQ_NS;
    int stamp_cnt = 0;
    int v = blockIdx.x;
    if (v >= units) return;""")
sub("                const char* asub = cur + kh * P2_SUB;", "                PQ_STAMP();\n                const char* asub = cur + kh * P2_SUB;")
sub("""                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);""", """                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
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
sub("""        __syncthreads();
        tile_m = n_tile_m; tile_n = n_tile_n;""", """        __syncthreads();
        PQ_STAMP();
        tile_m = n_tile_m; tile_n = n_tile_n;""")
sub("        sa = sa_n; sb = sb_n;\n    }\n}", """        sa = sa_n; sb = sb_n;
    }
    if (stamp_on) {
        __syncthreads();
        const unsigned* st = reinterpret_cast<const unsigned*>(smem + PQ_LDS);
        for (int i = tid; i < 8 * PQ_NS; i += NTB) pq_stamps_dev[(blockIdx.x == 0 ? 0 : 8 * PQ_NS) + i] = st[i];
    }
}
extern "C" int mh_exp_pq_stamps(unsigned* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pq_stamps_dev), sizeof(unsigned) * 2 * 8 * PQ_NS); }""")
sub("dim3(pq_grid(units_)), dim3(NTB), PQ_LDS, s, a, (int)units_,", "dim3(pq_grid(units_)), dim3(NTB), PQ_LDS + 8 * PQ_NS * 4, s, a, (int)units_,")
sub("hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);", "hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);")
open(sys.argv[1], "w").write(src)
