"""Register budget of the shipped HIP kernels, read from the compiler's own metadata (hipcc -save-temps): no kernel may
spill vector registers inside its matrix loop.  The step loops (rnn_h2.hip), the fused MLP tail, the heads and the
encoders must not spill at all (gru_rec_h2_kernel<128,256,true> carried 28 spilled VGPRs);
gemm_h2_kernel's 128x128 wave tile takes all 512 registers and parks eight accumulator registers in scratch for its
bias epilogue -- allowed only outside the span of its MFMAs.  CPU only: hipcc cross-compiles gfx950."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pepper_amd", "csrc")
STRICT = ["rnn_h2.hip", "mlp_h2.hip", "head.hip", "encoder.hip"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _asm(tmp_path, source):
    if not os.path.exists(_hipcc()):
        pytest.skip("hipcc not available")
    out = tmp_path / (source + ".o")
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", source, "-o", str(out), "-save-temps=obj"],
                   cwd=CSRC, check=True, capture_output=True)
    path = tmp_path / (source.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    return path.read_text()


def _kernels(asm):
    """{kernel symbol: (vgpr_count, vgpr_spill_count, sgpr_spill_count)} from the .amdhsa metadata block."""
    out = {}
    for block in asm.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        out[name] = (int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1)),
                     int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1)),
                     int(re.search(r"\.sgpr_spill_count:\s+(\d+)", block).group(1)))
    return out


def _body(asm, symbol):
    lines = asm.split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.startswith(symbol + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


@pytest.mark.parametrize("source", STRICT)
def test_no_vector_register_spills(tmp_path, source):
    kernels = _kernels(_asm(tmp_path, source))
    assert kernels, source
    spilled = {k: v for k, v in kernels.items() if v[1] != 0}
    assert not spilled, spilled
    if source == "rnn_h2.hip":
        # the instantiations the polish and variant models launch, by their mangled template arguments
        for needle in ("gru_rec_h2_kernelILi128ELi256ELb1ELi2ELb1E", "gru_rec_h2_kernelILi128ELi256ELb1ELi2ELb0E",
                       "gru_rec_h2_kernelILi128ELi16E", "gru_rec_h2_kernelILi128ELi128E",
                       "lstm_rec_h2_kernelILi256ELi512ELb1ELb1ELi2E", "lstm_rec_h2_kernelILi256ELi32ELb1ELb0ELi2E"):
            assert any(needle in k for k in kernels), needle


def test_step_loops_keep_their_scalars_in_scalar_registers(tmp_path):
    """Scalar registers spilled into vector lanes come back as v_readlane_b32 on the vector pipe, next to the MFMAs: r03's
    K = 768 LSTM loop carried 315 of them and 314 reloads per time step (every weight fragment's offset was `u`-dependent,
    so loop-invariant, so kept live).  The step loops the bench and the pipelines launch -- the LSTM's four forms, the GRU's
    three, the two split loops and the small-call GRU loop -- hold at most two spilled scalars, none of them reloaded inside
    the time loop (at most two reloads in the whole kernel: prologue / epilogue values)."""
    asm = _asm(tmp_path, "rnn_h2.hip")
    kernels = _kernels(asm)
    step_loops = [k for k in kernels if any(n in k for n in ("lstm_rec_h2_kernel", "gru_rec_h2_kernel", "lstm_rec_h2_split_kernel",
                                                               "gru_small_h2_kernel"))]
    assert len(step_loops) >= 10
    for name in step_loops:
        assert kernels[name][2] <= 2, (name, kernels[name])
        reloads = sum("v_readlane_b32" in ln for ln in _body(asm, name))
        # (the split loops and the fused heads use v_readlane for wave-uniform values on purpose: only the LSTM / GRU big loops
        # are held to the reload count)
        if "lstm_rec_h2_kernel" in name:
            assert reloads <= 2, (name, reloads)


def test_gemm_h2_spills_stay_out_of_the_matrix_loop(tmp_path):
    asm = _asm(tmp_path, "gemm_h2.hip")
    kernels = _kernels(asm)
    lines = asm.split("\n")
    for name, (vgprs, spills, _sgpr_spills) in kernels.items():
        assert spills <= 9, (name, spills)
        if spills == 0:
            continue
        start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        body = lines[start:end]
        mfma = [i for i, l in enumerate(body) if "v_mfma" in l]
        scratch = [i for i, l in enumerate(body) if "scratch_" in l]
        assert mfma and scratch
        inside = [i for i in scratch if mfma[0] < i < mfma[-1]]
        assert not inside, (name, len(inside))


def test_two_tile_workgroups_fit_a_cu(tmp_path):
    """tile_count_kernel lives on two workgroups per CU: with the tile's LDS at 82 052 B (one workgroup per CU) the same
    kernel took 2.70 ms instead of 1.93.  Its static LDS has to stay at or below half of the CU's 160 KB, and
    its registers at or below the 128 that let four waves share a SIMD."""
    asm = _asm(tmp_path, "encoder.hip")
    found = False
    for block in asm.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        if "tile_count_kernel" not in name:
            continue
        found = True
        lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", block).group(1))
        vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1))
        assert lds <= 160 * 1024 // 2, lds
        assert vgprs <= 128, vgprs
    assert found


def test_unit_split_step_loops_keep_their_weights_in_registers(tmp_path):
    """lstm_rec_h2_split_kernel (calls of at most 1024 windows) holds a wave's recurrent weights in registers for the whole step
    loop -- 128 VGPRs with eight members of 32 units, 256 with four of 64 -- and nothing may go to scratch.  The eight-member
    form has to stay at or below 256 registers: two of its workgroups (one wave per SIMD each) then share a CU, which is what
    lets two calls of 512 windows be on the GPU together (residency)."""
    asm = _asm(tmp_path, "rnn_h2.hip")
    kernels = _kernels(asm)
    eight = [v for k, v in kernels.items() if "lstm_rec_h2_split_kernelILi256ELi1E" in k]
    four = [v for k, v in kernels.items() if "lstm_rec_h2_split_kernelILi256ELi2E" in k]
    assert len(eight) == 1 and len(four) == 1
    assert eight[0][1] == 0 and four[0][1] == 0
    assert 128 < eight[0][0] <= 256, eight
    assert 256 < four[0][0] <= 512, four
    for needle in ("lstm_rec_h2_split_kernelILi256ELi1E", "lstm_rec_h2_split_kernelILi256ELi2E"):
        name = next(k for k in kernels if needle in k)
        lines = asm.split("\n")
        start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        body = lines[start:end]
        assert not any("scratch_" in l for l in body), needle
        # the exchange: sc1 on the 16-byte stores / loads, and no agent-scope fence (buffer_wbl2 / buffer_inv) anywhere
        assert sum("buffer_load_dwordx4" in l and "sc1" in l for l in body) == 8, needle
        assert any("buffer_store_dwordx4" in l and "sc1" in l for l in body), needle
        assert not any("buffer_wbl2" in l or "buffer_inv" in l for l in body), needle
