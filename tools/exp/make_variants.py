"""Build instrumented / ablated variants of libsdmi.so for kernel studies (not part of the product):

    python tools/exp/make_variants.py            # -> tools/exp/libsdmi_exp{1,2,3}.so
    SDMI_LIBPATH=tools/exp/libsdmi_exp1.so python tools/time_one.py 64 32 256 256 3

  exp1  igemm loaders skip their global loads (LDS stores, barriers and MFMAs kept)
  exp2  igemm MFMA waves skip the MFMAs (loaders unchanged)
  exp4  exp2 + no fragment reads (loaders + barriers only)
  dma   -DSDMI_IGEMM_DMA=1: big deep-K shapes go to the LDS-DMA kernels (igemm_dma_kernel)
  exp3  s_memtime stamps of workgroup 0 into the `workspace` argument (tools/exp/timeline.py)

The variants are produced by patching a COPY of slotdiffusion_amd/csrc/igemm.hip; if a pattern no
longer matches the current source the script stops."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, '..', '..')
CSRC = os.path.join(ROOT, 'slotdiffusion_amd', 'csrc')
sys.path.insert(0, ROOT)


def patch(src, old, new):
    assert old in src, f'pattern not found:\n{old[:200]}'
    return src.replace(old, new, 1)


def main():
    from slotdiffusion_amd.csrc import build
    build.build()
    src = open(os.path.join(CSRC, 'igemm.hip')).read()
    v = patch(src, "      if (ld_kt == 0) begin_tile();\n      if constexpr (TAPU || IS1X1) {",
              "      if (ld_kt == 0) begin_tile();\n#ifdef SDMI_EXP1\n      k0 += BK; "
              "if (++ld_kt == n_kt) { ld_kt = 0; ++ld_tile; } return;\n#endif\n      if constexpr (TAPU || IS1X1) {")
    v = patch(v, "          if constexpr (sizeof(T) == 2) {\n            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(",
              "#ifdef SDMI_EXP2\n          acc[i][j][0] += __uint_as_float(a4[0] ^ b4[0]);\n          continue;\n#endif\n"
              "          if constexpr (sizeof(T) == 2) {\n            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(")
    v = patch(v, "    read_frags(As, Bs, 0, fa[0], fb[0]);",
              "#ifdef SDMI_EXP4\n    continue;\n#endif\n    read_frags(As, Bs, 0, fa[0], fb[0]);")
    t = patch(src, """    for (int g = 0; g < total; g += 2) {
      store_tile(0, ra0, rb0);
      if (g + 2 < total) load_tile(ra0, rb0);
      __syncthreads();
      if (g + 1 < total) {
        store_tile(BUF_BYTES, ra1, rb1);
        if (g + 3 < total) load_tile(ra1, rb1);
        __syncthreads();
      }
    }""", """    long long* dbg = (long long*)p.workspace;
    const bool rec = dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
#define TS(slot, g_) do { if (rec && (g_) < 64) dbg[((g_) * 8 + (slot))] = __builtin_amdgcn_s_memtime(); } while (0)
    for (int g = 0; g < total; g += 2) {
      TS(0, g);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      TS(1, g);
      store_tile(0, ra0, rb0);
      TS(2, g);
      if (g + 2 < total) load_tile(ra0, rb0);
      TS(3, g);
      __syncthreads();
      if (g + 1 < total) {
        TS(0, g + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        TS(1, g + 1);
        store_tile(BUF_BYTES, ra1, rb1);
        TS(2, g + 1);
        if (g + 3 < total) load_tile(ra1, rb1);
        TS(3, g + 1);
        __syncthreads();
      }
    }""")
    t = patch(t, """  for (int t = 0; t < n_kt; ++t, ++g) {
    __syncthreads();                 // stage g & 1 holds this K tile""", """  long long* dbg = (long long*)p.workspace;
  const bool rec = dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
  for (int t = 0; t < n_kt; ++t, ++g) {
    if (rec && g < 64) dbg[g * 8 + 4] = __builtin_amdgcn_s_memtime();
    __syncthreads();                 // stage g & 1 holds this K tile
    if (rec && g < 64) dbg[g * 8 + 5] = __builtin_amdgcn_s_memtime();""")
    t = patch(t, """          }
        }
    }
  }

  wave_epilogue<TM, TN>(""", """          }
        }
    }
    if (rec && g < 64) dbg[g * 8 + 6] = __builtin_amdgcn_s_memtime();
  }
  wave_epilogue<TM, TN>(""")
    objs = [os.path.join(CSRC, '_build', o) for o in os.listdir(os.path.join(CSRC, '_build'))
            if o.endswith('.o') and o != 'igemm.o']
    for tag, text, defs in (('exp1', v, ['-DSDMI_EXP1']), ('exp2', v, ['-DSDMI_EXP2']), ('exp3', t, []),
                            ('exp4', v, ['-DSDMI_EXP2', '-DSDMI_EXP4']), ('dma', src, ['-DSDMI_IGEMM_DMA=1'])):
        cpy = f'/tmp/igemm_{tag}.hip'
        open(cpy, 'w').write(text)
        obj = f'/tmp/igemm_{tag}.o'
        subprocess.check_call([build.HIPCC] + build.FLAGS + defs + ['-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
                                                                    '-x', 'hip', '-c', cpy, '-o', obj])
        out = os.path.join(HERE, f'libsdmi_{tag}.so')
        subprocess.check_call([build.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
        print('built', out)


if __name__ == '__main__':
    main()
