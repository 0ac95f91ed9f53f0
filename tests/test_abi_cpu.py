"""The C-ABI library loads and exports every symbol include/ldpc_mi355x.h declares.
No compute happens here (there is no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest

import ldpcdecoders_jl_amd as ldpc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header="ldpc_mi355x.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ldpc_[a-z0-9_]+)\s*\(", txt)))


@pytest.mark.parametrize("experiments", [False, True])
def test_every_declared_symbol_is_exported(experiments):
    lib = ldpc._capi.lib(experiments)
    declared = _declared_functions()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ldpc_mi355x.h but not exported"
    assert sorted(ldpc._capi.EXPORTED_SYMBOLS) == declared
    assert lib.ldpc_abi_version() == 4 and lib.ldpc_build_target() == b"gfx950"
    # the test hooks live in a header of their own, outside the drop-in boundary
    hooks = _declared_functions("ldpc_mi355x_debug.h")
    assert sorted(ldpc._capi.DEBUG_SYMBOLS) == hooks and not set(hooks) & set(declared)
    for name in hooks:
        assert hasattr(lib, name)


def test_product_build_reads_no_environment_and_has_no_fault_injection():
    """The knobs of DESIGN.md's "Environment knobs" table exist in the experiments build only (host_env.hpp): the
    product library holds none of their names and does not import getenv at all."""
    import subprocess

    prod = open(ldpc._capi.LIB_PATH, "rb").read()
    exp = open(ldpc._capi.EXP_LIB_PATH, "rb").read()
    for knob in (b"LDPC_TEAM_INJECT_FAULT", b"LDPC_TEAM_CACHE_MIB", b"LDPC_DEFER_T0", b"LDPC_VMM_HINT_TIB", b"LDPC_NODE_MSG_LDS",
                 b"LDPC_BPOTS_FORCE_NODE", b"LDPC_WS_ALLOC"):
        assert knob not in prod, knob
        assert knob in exp, knob
    undefined = subprocess.run(["nm", "-D", "--undefined-only", ldpc._capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in undefined
    # the multi-GPU exchange binds RCCL at run time: no link-time dependency for single-GPU users
    needed = subprocess.run(["readelf", "-d", ldpc._capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()


def test_code_object_is_gfx950_only():
    data = open(ldpc._capi.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"nvptx"):
        assert other not in data


def test_argument_validation_happens_before_any_device_work():
    lib = ldpc._capi.lib()
    h = ctypes.c_void_p()
    colptr = np.array([0, 2, 2], dtype=np.int64)
    bad = np.array([1, 0], dtype=np.int64)
    st = lib.ldpc_bp_create(2, 2, 2, colptr.ctypes.data, bad.ctypes.data, 0.1, 5, None, ctypes.byref(h))
    assert st == 1 and b"ascending" in lib.ldpc_last_error() and not h.value
    st = lib.ldpc_bp_create(2, 2, 2, colptr.ctypes.data, np.array([0, 5], dtype=np.int64).ctypes.data,
                            0.1, 5, None, ctypes.byref(h))
    assert st == 1 and b"outside" in lib.ldpc_last_error()
    st = lib.ldpc_bp_create(2, 2, 3, colptr.ctypes.data, bad.ctypes.data, 0.1, 5, None, ctypes.byref(h))
    assert st == 1
    assert lib.ldpc_bp_decode_batch(None, 1, None, None, None, None, None) == 1
    assert lib.ldpc_bp_destroy(None) == 0
    # the multi-device constructor checks its own arguments first, too
    m = ctypes.c_void_p()
    good = np.array([0, 1], dtype=np.int64)
    devs = (ctypes.c_int32 * 2)(0, 0)
    assert lib.ldpc_bp_create_multi(0, devs, 0, 2, 2, 2, colptr.ctypes.data, good.ctypes.data, 0.1, 5, None, ctypes.byref(m)) == 1
    assert lib.ldpc_bp_create_multi(17, devs, 0, 2, 2, 2, colptr.ctypes.data, good.ctypes.data, 0.1, 5, None, ctypes.byref(m)) == 1
    assert lib.ldpc_bp_create_multi(2, devs, 7, 2, 2, 2, colptr.ctypes.data, good.ctypes.data, 0.1, 5, None, ctypes.byref(m)) == 1
    assert lib.ldpc_bp_decode_batch_multi(None, 1, None, None, None, None, None) == 1
    assert lib.ldpc_bp_decode_batch_multi_device(None, 1, None, None, None, None, None, None) == 1
    assert lib.ldpc_bp_destroy_multi(None) == 0 and not lib.ldpc_bp_multi_handle(None, 0)


def test_no_cpu_fallback():
    """Without a gfx950 device the product path must fail loudly, never compute on the CPU."""
    lib = ldpc._capi.lib()
    if lib.ldpc_device_count() > 0:
        pytest.skip("a GPU is present; the no-device behaviour is checked in the build container")
    H = ldpc.parity_check_matrix(96, 6, 3)
    with pytest.raises(ldpc.LdpcError) as ei:
        ldpc.BeliefPropagationDecoder(H, 0.01, 10)
    assert ei.value.status == 2
    # nothing in the product package reaches for the oracle
    pkg = os.path.join(ROOT, "ldpcdecoders.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".jl")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "bp_oracle" not in src, f


def test_constructor_type_checks():
    H = ldpc.parity_check_matrix(96, 6, 3)
    with pytest.raises(TypeError):
        ldpc.BeliefPropagationDecoder(H, 1, 10)        # per::Float64
    with pytest.raises(TypeError):
        ldpc.BeliefPropagationDecoder(H, 0.1, 10.0)    # max_iters::Int


def _build_driver(tmp_path):
    import subprocess

    exe = str(tmp_path / "abi_driver")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_driver.c"), "-o", exe,
                           ldpc._capi.LIB_PATH, "-Wl,-rpath," + os.path.dirname(ldpc._capi.LIB_PATH),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_header_is_valid_c_and_links_from_a_c_host(tmp_path):
    """include/ldpc_mi355x.h compiles as C99 and a plain-C program links against the library."""
    import subprocess

    exe = _build_driver(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
def test_c_host_decodes_on_the_gpu(tmp_path, gpu):
    import subprocess

    exe = _build_driver(tmp_path)
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0 and "gpu ok" in out.stdout, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
def test_c_host_multi_device_entries_equal_the_single_device_entry(tmp_path, gpu):
    """tests/abi_driver.c "multi": ldpc_bp_create_multi / ldpc_bp_decode_batch_multi[_device] from a plain-C host --
    ndev = 1 (the degenerate case: bit-equal to ldpc_bp_decode_batch_device), two logical devices on GPU 0 (shards
    exchanged with hipMemcpyPeerAsync) in the host and the root-device form, and ndev = 1 with the shard sent to
    itself through RCCL (ncclCommInitAll, grouped ncclSend / ncclRecv: the calls an 8-GPU run makes)."""
    import subprocess

    exe = _build_driver(tmp_path)
    out = subprocess.run([exe, "multi"], capture_output=True, text=True)
    assert out.returncode == 0 and "multi ok" in out.stdout, (out.returncode, out.stdout, out.stderr)


def _kernel_metadata(lib_path, tmp_path, tag):
    """{kernel symbol: {vgpr_count, agpr_count, private_segment_fixed_size}} and the unbundled code objects of a library."""
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    fat = str(tmp_path / f"{tag}_fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert starts, "no offload bundle in .hip_fatbin"
    notes, cos = "", []
    for k, a in enumerate(starts):   # one bundle per translation unit (pick_*.hip, ...)
        part, co = str(tmp_path / f"{tag}_bundle{k}.bin"), str(tmp_path / f"{tag}_dev{k}.co")
        open(part, "wb").write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"])
        notes += subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        cos.append(co)
    # one record per kernel in amdhsa.kernels: a YAML list item ("  - .agpr_count: ...") whose keys come in alphabetical order
    info, rec = {}, None
    for line in notes.splitlines():
        if re.match(r"  - \.", line):              # (the items of .args and the like sit deeper)
            rec = {}
        m = re.match(r"\s*(?:- )?\.(name|vgpr_count|agpr_count|private_segment_fixed_size):\s+(\S+)", line)
        if not m or rec is None:
            continue
        if m.group(1) == "name":
            if m.group(2).startswith("_Z"):
                info[m.group(2)] = rec
        else:
            rec[m.group(1)] = int(m.group(2))
    return info, cos


def test_hot_kernels_keep_their_register_budget(tmp_path):
    """Three workgroups of 8 waves per CU need <= 80 VGPRs (512 / 6 waves per SIMD) and no scratch: the
    tile kernel's C3 instantiation and the LDS kernel's C2 instantiation sit right at that edge, and an
    innocent edit has cost a workgroup per CU (-20 % throughput) more than once.  Read from the code
    object's metadata in the built library."""
    import shutil
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    if not (shutil.which("objcopy") and os.path.exists(f"{llvm}/clang-offload-bundler") and os.path.exists(f"{llvm}/llvm-readelf")):
        pytest.skip("no code-object tools")
    info, _ = _kernel_metadata(ldpc._capi.LIB_PATH, tmp_path, "prod")
    hot = [k for k in info if k.startswith("_ZN4ldpc14bp_tile_kernelILi8ELi4ELb0ELi512ELb0EEE")
           or k.startswith("_ZN4ldpc13bp_lds_kernelILi8ELi4ELb0ELi512EEE")]
    assert len(hot) == 2, sorted(info)[:5]
    for k in hot:
        assert info[k]["vgpr_count"] <= 80 and info[k]["private_segment_fixed_size"] == 0, (k, info[k])
    # every rows-on-chip instantiation of the team kernel -- check degree 6 ... 10 x bit degree 3 ... 5, with and without
    # LLRs --: one 8-wave workgroup per CU (156 KB of LDS), not a byte of scratch memory
    rows = [k for k in info if re.match(r"_ZN4ldpc14bp_team_kernelILi\d+ELi\d+ELb[01]ELi512ELb0ELb1ELi0ELb0EEE", k)]
    assert len(rows) == 30, [k for k in info if "bp_team_kernel" in k][:8]
    for k in rows:
        assert info[k]["private_segment_fixed_size"] == 0 and info[k]["vgpr_count"] <= 256, (k, info[k])
    # the instantiations for IRREGULAR graphs (whole checks in LDS, by register bucket): one workgroup per CU as well, no scratch
    irr = [k for k in info if re.match(r"_ZN4ldpc14bp_team_kernelILi\d+ELi\d+ELb[01]ELi512ELb0ELb0ELi0ELb1EEE", k)]
    assert len(irr) == 8, [k for k in info if "bp_team_kernel" in k][:8]
    for k in irr:
        assert info[k]["private_segment_fixed_size"] == 0 and info[k]["vgpr_count"] <= 256, (k, info[k])
    # ... and the ones that also keep rows in the top 64 registers of every wave (bp_team_kernels.hpp "Rows in
    # REGISTERS"; the headline instantiation is one of them): 256 registers a lane, no accumulator registers (the
    # allocator would park values of its own in them), no scratch, and nothing but the two accessors (v_mov_b32 from / to
    # v192 | v193 under s_set_gpr_idx_on) may touch v192 and up
    for tag, path in (("prod", ldpc._capi.LIB_PATH), ("exp", ldpc._capi.EXP_LIB_PATH)):
        xinfo, cos = _kernel_metadata(path, tmp_path, tag + "2")
        regs = [k for k in xinfo if re.match(r"_ZN4ldpc14bp_team_kernelILi\d+ELi\d+ELb[01]ELi512ELb0ELb1ELi32ELb0EEE", k)]
        assert len(regs) == 30
        for k in regs:
            assert xinfo[k]["private_segment_fixed_size"] == 0 and xinfo[k]["agpr_count"] == 0 and xinfo[k]["vgpr_count"] == 256, (k, xinfo[k])
        top = re.compile(r"\bv(19[2-9]|2[0-4]\d|25[0-5])\b|\bv\[\d+:(19[2-9]|2[0-4]\d|25[0-5])\]")
        ok_form = re.compile(r"^\s*v_mov_b32_e32 (v\d+, v19[23]|v19[23], v\d+)\b")
        checked = idx_pairs = 0
        for co in cos:
            dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            if "ELb0ELb1ELi32ELb0EEE" not in dis:
                continue
            cur = None
            # M0: s_set_gpr_idx_on writes it, so every accessor saves it into a scalar register first and puts it back after
            # s_set_gpr_idx_off, with nothing in between but the s_nop and the two moves (bp_team_kernels.hpp)
            prev, inside, saved = "", None, None
            for line in dis.splitlines():
                m = re.match(r"[0-9a-f]+ <(\S+)>:", line)
                if m:
                    assert inside is None, (cur, "index mode left on at the end of a function")
                    cur = m.group(1)
                    prev = ""
                    continue
                if cur not in regs:
                    continue
                ins = line.split("//")[0].strip()
                if not ins:
                    continue
                if top.search(ins):
                    assert ok_form.match(line), (cur, line)
                    assert inside is not None, (cur, line, "a top register touched outside the index mode")
                    checked += 1
                if ins.startswith("s_set_gpr_idx_on"):
                    ms = re.match(r"s_mov_b32 (s\d+), m0$", prev)
                    assert ms, (cur, prev, ins, "M0 is not saved right before s_set_gpr_idx_on")
                    saved, inside = ms.group(1), []
                    idx_pairs += 1
                elif ins.startswith("s_set_gpr_idx_off"):
                    assert inside is not None and len(inside) == 3 and inside[0].startswith("s_nop") and \
                        all(q.startswith("v_mov_b32") for q in inside[1:]), (cur, inside)
                    inside = None
                elif inside is not None:
                    inside.append(ins)
                elif saved is not None:
                    assert ins == f"s_mov_b32 m0, {saved}", (cur, ins, "M0 is not restored right after s_set_gpr_idx_off")
                    saved = None
                prev = ins
        assert checked > 100 and idx_pairs * 2 == checked, (checked, idx_pairs)


def test_wait_limit_is_a_per_process_setting(ldpc):
    """ldpc_set_wait_limit_ms / ldpc_get_wait_limit_ms (host_wait.hpp): the bound of every host-side wait; default ten
    minutes, 0 = unbounded, negative rejected; needs no device."""
    for exp_build in (False, True):
        L = ldpc._capi.lib(exp_build)
        assert L.ldpc_get_wait_limit_ms() == 600000
        assert L.ldpc_set_wait_limit_ms(-1) == 1 and b"negative" in L.ldpc_last_error()
        assert L.ldpc_set_wait_limit_ms(0) == 0 and L.ldpc_get_wait_limit_ms() == 0
        assert L.ldpc_set_wait_limit_ms(12345) == 0 and L.ldpc_get_wait_limit_ms() == 12345
        assert L.ldpc_set_wait_limit_ms(600000) == 0
