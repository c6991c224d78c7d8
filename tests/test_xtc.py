"""CPU tier: the XTC decoder in libmkamd.so (host code, no GPU) against what the REAL reference reader decodes from
the same files (tests/golden/xtc/*, made by tests/golden/make_golden_xtc.py).  Integer/bit work: the bar is
bit-exact float32 coordinates, box vectors, times and steps."""
import os

import numpy as np
import pytest

from moleculekit_amd import xtc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xtc")
FILES = ["mol", "aladipep", "3ptb_traj_head", "4rws_head"]


def _fn(name):
    return os.path.join(HERE, name + ".xtc")


@pytest.mark.parametrize("name", FILES)
@pytest.mark.parametrize("nthreads", [1, 0])
def test_read_xtc_bit_exact(name, nthreads):
    g = np.load(os.path.join(HERE, name + "_decoded.npz"))
    coords, box, time, step = xtc.read_xtc(_fn(name), nthreads=nthreads)
    st = int(g["stride"])
    assert coords.dtype == np.float32 and coords.shape[0] == int(g["natoms"]) and coords.flags["C_CONTIGUOUS"]
    assert np.array_equal(coords[::st].view(np.uint32), g["coords"].view(np.uint32))
    assert int(coords.view(np.uint32).astype(np.uint64).sum()) == int(g["bitsum"])      # every atom, not just the stored ones
    assert np.array_equal(box, g["box"]) and np.array_equal(time, g["time"]) and np.array_equal(step, g["step"])
    assert xtc.get_xtc_natoms(_fn(name)) == coords.shape[0] and xtc.get_xtc_nframes(_fn(name)) == coords.shape[2]


@pytest.mark.parametrize("name", FILES)
def test_read_xtc_frames_selection(name):
    g = np.load(os.path.join(HERE, name + "_decoded.npz"))
    c, b, t, s = xtc.read_xtc_frames(_fn(name), g["sel"])
    st = int(g["stride"])
    assert np.array_equal(c[::st], g["sel_coords"]) and np.array_equal(b, g["sel_box"])
    assert np.array_equal(t, g["sel_time"]) and np.array_equal(s, g["sel_step"])
    # repeats and arbitrary order are columns of the full read
    full = xtc.read_xtc(_fn(name))[0]
    F = full.shape[2]
    sel = np.array([F - 1, 0, F - 1, F // 2])
    assert np.array_equal(xtc.read_xtc_frames(_fn(name), sel)[0], full[:, :, sel])


def test_xtcread_units_and_box():
    """readers.XTCread (readers.py:1846-1860): nm -> A, ps -> fs, box vectors -> lengths / angles."""
    raw_c, raw_b, raw_t, raw_s = xtc.read_xtc(_fn("aladipep"))
    tr = xtc.XTCread(_fn("aladipep"))
    assert np.array_equal(tr.coords, raw_c * np.float32(10.0)) and tr.coords.dtype == np.float32
    assert np.allclose(tr.time, raw_t.astype(np.float64) * 1e3) and np.array_equal(tr.step, raw_s)
    assert tr.box.shape == (3, raw_c.shape[2]) and np.allclose(tr.box[0], raw_b[0, 0] * 10.0, rtol=1e-6)
    assert np.allclose(tr.boxangles, 90.0)
    one = xtc.XTCread(_fn("aladipep"), frame=3)
    assert one.coords.shape[2] == 1 and np.array_equal(one.coords[:, :, 0], tr.coords[:, :, 3])
    la, lb, lc, al, be, ga = xtc.box_vectors_to_lengths_and_angles(np.array([2.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0, 1.0, 1.0]))
    assert la == 2.0 and lb == 1.0 and np.isclose(lc, np.sqrt(2)) and np.isclose(al, 45) and np.isclose(be, 90) and np.isclose(ga, 90)


def test_truncated_and_bad_files(tmp_path):
    buf = open(_fn("aladipep"), "rb").read()
    cut = tmp_path / "cut.xtc"
    cut.write_bytes(buf[:len(buf) - 1000])                       # last record incomplete: it is not a frame
    nf = xtc.get_xtc_nframes(str(cut))
    assert nf == 19
    assert np.array_equal(xtc.read_xtc(str(cut))[0], xtc.read_xtc(_fn("aladipep"))[0][:, :, :19])
    with pytest.raises(ValueError, match="out of range"):
        xtc.read_xtc_frames(str(cut), [19])
    bad = tmp_path / "bad.xtc"
    bad.write_bytes(b"not an xtc file at all, just some text that is long enough to hold a header" * 2)
    with pytest.raises(ValueError, match="XTC"):
        xtc.read_xtc(str(bad))
    with pytest.raises(ValueError, match="cannot open"):
        xtc.get_xtc_natoms(str(tmp_path / "missing.xtc"))
    corrupt = bytearray(buf)
    corrupt[200:260] = b"\\xff" * 60                              # garbage inside the first frame's bit stream
    cf = tmp_path / "corrupt.xtc"
    cf.write_bytes(bytes(corrupt))
    try:                                                         # must not crash; either an error or finite garbage
        c = xtc.read_xtc(str(cf))[0]
        assert c.shape[0] == 688
    except ValueError:
        pass


def test_frame_index_is_remembered_per_file_state_not_per_path(tmp_path):
    """The reader keeps the frame index of the file it read last (a streaming caller asks chunk after chunk); a file
    REPLACED under the same path -- other frame count, other atom count -- must be indexed afresh."""
    import shutil
    a, b = _fn("aladipep"), _fn("3ptb_traj_head")
    p = str(tmp_path / "t.xtc")
    shutil.copyfile(a, p)
    ca1 = xtc.read_xtc(p)[0]
    ca2 = xtc.read_xtc_frames(p, [0])[0]                     # second call on the same file: the remembered index
    assert np.array_equal(ca1[:, :, :1], ca2)
    shutil.copyfile(b, p)
    cb = xtc.read_xtc(p)[0]
    assert np.array_equal(cb, xtc.read_xtc(b)[0]) and np.array_equal(ca1, xtc.read_xtc(a)[0])
    assert (xtc.get_xtc_natoms(p), xtc.get_xtc_nframes(p)) == (xtc.get_xtc_natoms(b), xtc.get_xtc_nframes(b))


def test_threaded_decode_into_a_fresh_big_array_equals_the_serial_one(tmp_path):
    """Results of 8 MB and more are pre-touched by the decode threads (contiguous slices, zeros) before the frames are
    written: a file long enough to take that path -- the 3PTB fixture's frames repeated, XTC frames are self-contained
    records -- decoded by 1, 4 and the default number of threads, and an index list in shuffled order."""
    src = open(_fn("3ptb_traj_head"), "rb").read()
    p = str(tmp_path / "long.xtc")
    with open(p, "wb") as f:
        for _ in range(40):
            f.write(src)
    serial = xtc.read_xtc(p, nthreads=1)
    assert serial[0].nbytes >= (8 << 20)
    for nt in (4, 0):
        got = xtc.read_xtc(p, nthreads=nt)
        assert all(np.array_equal(a, b) for a, b in zip(got, serial))
    order = np.random.default_rng(0).permutation(serial[0].shape[2])
    picked = xtc.read_xtc_frames(p, order, nthreads=4)
    assert np.array_equal(picked[0], serial[0][:, :, order]) and np.array_equal(picked[3], serial[3][order])


@pytest.mark.parametrize("name", ["small9", "mixed_radix", "negative_origin", "per_axis_fields"])
def test_write_xtc_is_read_back_like_the_reference_reads_it(name, tmp_path):
    """Round 4: moleculekit_amd.xtc.write_xtc (the signature of moleculekit.xtc.write_xtc, xtc.pyx:86-97).  The golden file holds
    what the REAL reference reader decoded from files this writer produced (tests/golden/make_golden_xtc_writer.py: up to
    nine atoms stored as floats; the mixed-radix form; negative minima; ranges beyond 24 bits = per-axis bit fields): the
    file written here again must decode -- with this package's decoder -- to exactly those arrays, and stay within half a
    quantum (0.5 / 1000 nm) of what was written, as far as float32 resolves it."""
    g = np.load(os.path.join(os.path.dirname(HERE), "xtc_writer_cases.npz"))
    x, box, t, st = (g[f"{name}_{k}"] for k in ("coords", "box", "time", "step"))
    fn = str(tmp_path / "w.xtc")
    xtc.write_xtc(fn, x, box, t, st)
    assert xtc.get_xtc_natoms(fn) == x.shape[0] and xtc.get_xtc_nframes(fn) == x.shape[2]
    c, b, tt, ss = xtc.read_xtc(fn)
    assert np.array_equal(c.view(np.uint32), g[f"{name}_ref_coords"].view(np.uint32))
    assert np.array_equal(b, g[f"{name}_ref_box"]) and np.array_equal(tt, g[f"{name}_ref_time"]) and np.array_equal(ss, g[f"{name}_ref_step"])
    assert np.array_equal(b, box) and np.array_equal(tt, t) and np.array_equal(ss, st)
    assert np.abs(c - x).max() <= 0.5005e-3 + np.abs(x).max() * 2.0 ** -22
    with pytest.raises(ValueError):
        xtc.write_xtc(fn, x[:, :2], box, t, st)


def test_write_then_read_round_trip_property():
    """Any trajectory written by write_xtc comes back from read_xtc as the quantised coordinates -- int(x * 1000 +- 0.5) / 1000
    in float32 arithmetic, what the format stores (xdrfile.cpp:606-624 / :975-983) -- for random atom counts around the
    9-atom switch to plain floats, frame counts, box sizes from a ligand's to beyond 24 bits of range, mixed signs."""
    hypothesis = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    import tempfile

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 70), st.integers(1, 4), st.sampled_from([0.3, 2.5, 6.69, 40.0, 3000.0, 20000.0]), st.integers(0, 2 ** 31 - 1))
    def prop(n, f, L, seed):
        rng = np.random.default_rng(seed)
        x = (rng.uniform(-0.5, 1.0, size=(n, 3, f)) * L).astype(np.float32)
        box = np.zeros((3, 3, f), np.float32); box[0, 0] = box[1, 1] = box[2, 2] = L
        t, s = rng.uniform(0, 100, f).astype(np.float32), rng.integers(0, 10 ** 6, f)
        with tempfile.TemporaryDirectory() as d:
            fn = os.path.join(d, "p.xtc")
            xtc.write_xtc(fn, x, box, t, s)
            c, b, tt, ss = xtc.read_xtc(fn, nthreads=1)
            c2 = xtc.read_xtc_frames(fn, np.arange(f)[::-1])[0]
        lf = x * np.float32(1000)
        q = np.where(lf >= 0, lf + np.float32(0.5), lf - np.float32(0.5)).astype(np.int64)
        want = x if n <= 9 else q.astype(np.float32) * np.float32(1.0 / np.float32(1000.0))
        assert np.array_equal(c, want) and np.array_equal(c2, want[:, :, ::-1])
        assert np.array_equal(b, box) and np.array_equal(tt, t) and np.array_equal(ss, s.astype(np.int32))

    prop()


def _device_decode_emulated(fn, sel=None, scale=10.0):
    """The host half of the device decoder (headers, record bytes: libmkamd.so, no GPU involved) + the two kernels of
    csrc/xtc_gpu.h run by the host SIMT emulation (tests/emu): -> (xyz [n, natoms, 3], status [n])."""
    import ctypes
    from tests import emu_build
    from moleculekit_amd import _lib
    na, nf = xtc.get_xtc_natoms(fn), xtc.get_xtc_nframes(fn)
    sel = np.arange(nf, dtype=np.int64) if sel is None else np.asarray(sel, dtype=np.int64)
    desc, lo, hi, box, t, st = xtc.chunk_desc(fn, sel, na)
    raw = np.zeros(hi - lo + xtc.XTC_PAD, np.uint8)
    _lib._check(_lib.load().mkamd_xtc_copy_bytes(xtc._path(fn), lo, hi, raw.ctypes.data_as(ctypes.c_void_p), 1))
    assert np.array_equal(raw[:hi - lo], np.fromfile(fn, np.uint8, count=hi - lo, offset=lo))
    return emu_build.xtc_decode(raw, desc, na, scale) + (desc,)


def test_device_decoder_kernels_emulated_bit_exact_with_the_host_decoder(tmp_path):
    """csrc/xtc_gpu.h on the CPU tier: the walk (a lane per frame: flag and run bits out of refilled windows, group records)
    and the expansion (a thread per group) compiled for the host and run lane by lane (tests/emu) -- every reference-held
    trajectory (runs of small atoms, the water swap, steps of the small-number table), frames in any order, more frames than a
    wave, per-axis bit fields, <= 9 atoms, streams of many windows: the host decoder's coordinates bit for bit (itself pinned
    against the real reference reader above), scaled with the same float32 multiply.  Numbers of more than 64 bits and damaged
    streams come back as statuses, nothing is written out of bounds (the emulator's buffers are exact-size numpy arrays)."""
    files = [os.path.join(HERE, n + ".xtc") for n in ("mol", "aladipep", "3ptb_traj_head", "4rws_head")]
    rng = np.random.default_rng(21)
    for name, N, F, L in (("syn", 1500, 70, 6.69), ("small", 7, 4, 2.0), ("wide", 50, 3, 30000.0), ("tiny_box", 200, 66, 0.8)):
        x = (rng.uniform(-0.3, 1.0, size=(N, 3, F)) * L).astype(np.float32)
        bv = np.zeros((3, 3, F), np.float32); bv[0, 0] = bv[1, 1] = bv[2, 2] = L
        fn = str(tmp_path / (name + ".xtc"))
        xtc.write_xtc(fn, x, bv, np.arange(F, dtype=np.float32), np.arange(F))
        files.append(fn)
    for fn in files:
        F = xtc.get_xtc_nframes(fn)
        for sel in (None, np.arange(F)[::-1][: max(1, F // 2)], np.array([F - 1, 0, F - 1])):
            c = (xtc.read_xtc(fn) if sel is None else xtc.read_xtc_frames(fn, sel))[0]
            want = np.ascontiguousarray(np.transpose(c, (2, 0, 1))) * np.float32(10.0)
            got, st, _ = _device_decode_emulated(fn, sel)
            assert not st.any(), (os.path.basename(fn), st)
            assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), os.path.basename(fn)
    # 66..72-bit mixed-radix numbers: refused (status 2), and the headers say so beforehand
    x = (rng.uniform(-0.5, 1.0, size=(40, 3, 2)) * 3000.0).astype(np.float32)
    bv = np.zeros((3, 3, 2), np.float32); bv[0, 0] = bv[1, 1] = bv[2, 2] = 3000.0
    fn = str(tmp_path / "huge.xtc")
    xtc.write_xtc(fn, x, bv, np.zeros(2, np.float32), np.arange(2))
    got, st, desc = _device_decode_emulated(fn)
    assert list(st) == [2, 2] and np.isnan(got).all() and not xtc.device_decodable(desc, 40)
    assert xtc.device_decodable(_device_decode_emulated(files[2])[2], xtc.get_xtc_natoms(files[2]))
    # damaged streams: an impossible small-number index; a byte count that ends the stream early; random bit flips never
    # write outside the frame's rows and either flag the frame or decode like the host decoder does
    src = files[2]
    na = xtc.get_xtc_natoms(src)
    desc0, lo, hi, _, _, _ = xtc.chunk_desc(src, np.arange(3), na)
    d = desc0.view(xtc.DESC_DTYPE).reshape(-1)
    blob = bytearray(open(src, "rb").read())
    off = int(d["data_off"][1]) + lo - 8                                                   # the header's smallidx word
    assert int.from_bytes(blob[off:off + 4], "big") == int(d["smallidx"][1])
    bad = bytearray(blob); bad[off:off + 4] = (200).to_bytes(4, "big")
    fn = str(tmp_path / "bad_idx.xtc"); open(fn, "wb").write(bytes(bad))
    got, st, _ = _device_decode_emulated(fn, np.arange(3))
    assert list(st) == [0, 1, 0] and np.isnan(got[1]).all() and not np.isnan(got[0]).any() and not np.isnan(got[2]).any()
    raw = np.zeros(hi - lo + xtc.XTC_PAD, np.uint8); raw[:hi - lo] = np.frombuffer(bytes(blob[lo:hi]), np.uint8)
    from tests import emu_build
    short = desc0.copy(); short.view(xtc.DESC_DTYPE).reshape(-1)["nbytes"][1] //= 2           # the stream "ends" half way
    got, st = emu_build.xtc_decode(raw, short, na, 10.0)
    assert list(st) == [0, 1, 0]
    ok = emu_build.xtc_decode(raw, desc0, na, 10.0)[0]
    for trial in range(6):
        r2 = raw.copy()
        pos = rng.integers(int(d["data_off"][1]), int(d["data_off"][1]) + int(d["nbytes"][1]), size=4)
        r2[pos] ^= rng.integers(1, 256, size=4).astype(np.uint8)
        got, st = emu_build.xtc_decode(r2, desc0, na, 10.0)
        assert st[0] == 0 and st[2] == 0 and np.array_equal(got[0], ok[0]) and np.array_equal(got[2], ok[2])
        assert st[1] in (0, 1)


@pytest.mark.parametrize("name", ["metricdistance_traj.xtc"])
def test_headers_parsed_from_a_copy_of_the_bytes_equal_the_ones_parsed_from_the_file(name, tmp_path):
    """Round 6: a streaming reader copies the records' bytes first (their range from the frame index alone: mkamd_xtc_byte_range; the copy by
    pread in threads) and parses the headers out of the copy (mkamd_xtc_chunk_desc_mem): the same descriptors, range, boxes, times and steps
    as mkamd_xtc_chunk_desc reads from the file -- for contiguous chunks, single frames, reversed and scattered selections, the file's last
    frame; and the refusals (a copy that does not hold a selected frame, a range outside the file)."""
    import ctypes
    from moleculekit_amd import _lib
    fn = os.path.join(os.path.dirname(__file__), "golden", "xtc", name)
    na, nf = xtc.get_xtc_natoms(fn), xtc.get_xtc_nframes(fn)
    rng = np.random.default_rng(3)
    sels = [np.arange(nf), np.arange(min(3, nf)), np.array([nf - 1]), np.arange(nf)[::-1], np.sort(rng.choice(nf, max(1, nf // 2), replace=False)),
            np.array([0, nf - 1])]
    for sel in sels:
        want = xtc.chunk_desc(fn, sel, na)
        lo, hi = xtc.byte_range(fn, sel, na)
        assert lo == want[1] and hi >= want[2]
        for threads in (1, 3):
            raw = np.full(hi - lo + 7, 0xAB, np.uint8)
            _lib._check(_lib.load().mkamd_xtc_copy_bytes(xtc._path(fn), lo, hi, raw.ctypes.data_as(ctypes.c_void_p), threads))
            assert np.array_equal(raw[:hi - lo], np.fromfile(fn, np.uint8, count=hi - lo, offset=lo)) and (raw[hi - lo:] == 0xAB).all()
        got = xtc.chunk_desc_mem(fn, sel, na, raw.ctypes.data, lo, hi)
        assert got[1] == want[1] and got[2] == want[2]
        for a, b in zip(got, want):
            if isinstance(a, np.ndarray):
                assert np.array_equal(a, b)
    if nf >= 2:
        lo1, hi1 = xtc.byte_range(fn, np.array([1]), na)
        raw = np.fromfile(fn, np.uint8, count=hi1 - lo1, offset=lo1)
        with pytest.raises(Exception, match="outside the bytes"):
            xtc.chunk_desc_mem(fn, np.array([0, 1]), na, raw.ctypes.data, lo1, hi1)
    with pytest.raises(Exception, match="outside the file"):
        xtc.chunk_desc_mem(fn, np.array([0]), na, raw.ctypes.data, 0, os.path.getsize(fn) + 1)
    with pytest.raises(Exception, match="outside the file"):
        _lib._check(_lib.load().mkamd_xtc_copy_bytes(xtc._path(fn), 0, os.path.getsize(fn) + 1, raw.ctypes.data_as(ctypes.c_void_p), 1))
