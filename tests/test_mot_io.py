"""MOT I/O + motcpp_eval-compatible command line (reference: src/data/mot17_dataset.cpp:12-241,
include/motcpp/utils/mot_format.hpp:23-79, tools/motcpp_eval.cpp:19-468). The reader/writer are host-only C++ and are
checked on CPU against this file's own reading of the same fixtures; the command line is run on the GPU box and its
result files are compared byte for byte with the oracle's tables pushed through a Python mirror of the writer."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import mot17, orclib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "test_mot_io")
EVAL = os.path.join(ROOT, "motcpp_amd", "bin", "motcpp_eval")


def build():
    from motcpp_amd import _lib
    if not (os.path.exists(_lib.HIP_LIB) and os.path.exists(_lib.HOST_LIB) and os.path.exists(EVAL)):
        _lib.build()
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_mot_io.cpp")
    if not os.path.exists(BIN) or os.path.getmtime(src) > os.path.getmtime(BIN) or os.path.getmtime(_lib.HOST_LIB) > os.path.getmtime(BIN):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-o", BIN,
                               "-L", _lib.LIBDIR, "-lmotcpp", "-lmotcpp_hip", "-Wl,-rpath," + _lib.LIBDIR])
    return BIN


def make_root(tmp, gt_frames=None, seqinfo=True):
    """<tmp>/train/<seq>/det/det.txt from the gzipped fixtures (+ seqinfo.ini, optional gt.txt with frames 1..gt_frames)."""
    root = os.path.join(tmp, "train")
    for seq in mot17.SEQS:
        d = os.path.join(root, seq, "det")
        os.makedirs(d)
        with gzip.open(os.path.join(ROOT, "tests", "golden", "MOT17-mini", seq + ".det.txt.gz"), "rb") as src, \
                open(os.path.join(d, "det.txt"), "wb") as dst:
            shutil.copyfileobj(src, dst)
        if seqinfo:
            with open(os.path.join(root, seq, "seqinfo.ini"), "w") as f:
                f.write("[Sequence]\nname=%s\nframeRate=25\nimWidth=1920\nimHeight=1080\n" % seq)
        if gt_frames:
            os.makedirs(os.path.join(root, seq, "gt"))
            with open(os.path.join(root, seq, "gt", "gt.txt"), "w") as f:
                for fr in range(1, gt_frames[seq] + 1):
                    f.write("%d,1,10,10,20,40,1,1,1\n" % fr)
    open(os.path.join(root, "not-a-sequence.txt"), "w").close()
    return root


def mot_lines(table, frame):
    """Python mirror of convert_to_mot_format + write_mot_results (mot_format.hpp:23-79)."""
    out = []
    for r in np.asarray(table, np.float32):
        w, h = np.float32(r[2] - r[0]), np.float32(r[3] - r[1])
        out.append("%d,%d,%d,%d,%d,%d,%.6f,-1,-1,-1\n" % (frame, int(r[4]), int(r[0]), int(r[1]), int(w), int(h), float(r[5])))
    return out


def test_reader_matches_fixture_reading(tmp_path):
    root = make_root(str(tmp_path))
    out = subprocess.run([build(), "index", root], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("SEQ ")]
    assert [l.split()[1] for l in lines] == sorted(mot17.SEQS)  # sorted by name, the stray file is ignored
    for l in lines:
        kv = dict(p.split("=") for p in l.split()[2:])
        seq = l.split()[1]
        frames = mot17.load(seq)
        nonempty = [i + 1 for i, f in enumerate(frames) if f.shape[0]]
        assert int(kv["fps"]) == 25 and kv["size"] == "1920x1080" and kv["det"] == "det.txt"
        assert int(kv["frames"]) == len(nonempty) and int(kv["first"]) == nonempty[0] and int(kv["last"]) == nonempty[-1]
        assert int(kv["rows"]) == sum(f.shape[0] for f in frames) and int(kv["max_rows"]) == max(f.shape[0] for f in frames)
        want = sum(float((f.astype(np.float64) * np.arange(1, 7)).sum()) for f in frames)
        assert abs(float(kv["sum"]) - want) <= 1e-6 * abs(want)  # same float32 boxes (x2 = x1 + w in float), file order irrelevant here
    assert "NOTFOUND 1" in out.stdout


def test_reader_defaults_and_pregenerated_format(tmp_path):
    root = make_root(str(tmp_path), seqinfo=False)
    # pre-generated detections: <det_emb_root>/<model>/dets/MOT17-02.txt, whitespace separated "frame x1 y1 x2 y2 conf cls"
    dets_dir = os.path.join(str(tmp_path), "yolox", "yolox", "dets")
    os.makedirs(dets_dir)
    with open(os.path.join(dets_dir, "MOT17-02.txt"), "w") as f:
        f.write("# comment\n1 10 20 30 60 0.9 0\n1 100 20 130 60 0.8 2\n3 5 5 15 25 0.7 0\nshort 1 2\n")
    os.makedirs(os.path.join(root, "MOT17-04-FRCNN", "img1"))  # indexed through its image directory (the reference's rule)
    open(os.path.join(root, "MOT17-04-FRCNN", "img1", "000007.jpg"), "w").close()
    out = subprocess.run([build(), "index", root, os.path.join(str(tmp_path), "yolox"), "yolox"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = {l.split()[1]: dict(p.split("=") for p in l.split()[2:]) for l in out.stdout.splitlines() if l.startswith("SEQ ")}
    a = lines["MOT17-02-FRCNN"]
    assert a["fps"] == "30" and a["det"] == "MOT17-02.txt" and a["frames"] == "2" and a["rows"] == "3" and a["last"] == "3"
    assert lines["MOT17-04-FRCNN"]["frames"] == "0"  # no file for it under dets/: empty map, not an error


def test_writer_format(tmp_path):
    path = os.path.join(str(tmp_path), "sub", "dir", "out.txt")  # parent directories are created
    assert subprocess.run([build(), "write", path], timeout=60).returncode == 0
    t = np.array([[100.7, 50.2, 180.9, 260.4, 7, 0.912345678, 0, 3], [-5.5, 10.99, 20.25, 99.999, 12, 0.5, 1, 0],
                  [1919.6, 1000.4, 1925.1, 1085.9, 3, 1.0, 0, 1]], np.float32)
    assert open(path).read() == "".join(mot_lines(t, 42) + mot_lines(t, 43))  # the second call appends
    assert open(path).readline() == "42,7,100,50,80,210,0.912346,-1,-1,-1\n"  # truncation, 6 decimals


def test_cli_usage_and_unknown_method(tmp_path):
    build()
    assert subprocess.run([EVAL], capture_output=True, timeout=60).returncode == 1
    root = make_root(str(tmp_path))
    out = subprocess.run([EVAL, root, os.path.join(str(tmp_path), "res"), "fairmot"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "unknown tracking method" in out.stderr
    out = subprocess.run([EVAL, os.path.join(str(tmp_path), "nope"), os.path.join(str(tmp_path), "res")], capture_output=True, text=True,
                         timeout=60)
    assert out.returncode == 1 and "does not exist" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("method,kind,params", [("sort", orclib.SORT, [0.3, 1, 50, 3, 0.3]),
                                                ("bytetrack", orclib.BYTETRACK, [0.1, 0.45, 0.8, 30, 25, 30, 50])])
def test_cli_results_match_oracle(tmp_path, orc, method, kind, params):
    build()
    root, res = make_root(str(tmp_path)), os.path.join(str(tmp_path), "results")
    out = subprocess.run([EVAL, root, res, method], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    for seq in mot17.SEQS:
        trk = orc.tracker(kind, params)
        want = []
        for f, d in enumerate(mot17.load(seq), start=1):
            if d.shape[0] == 0:
                continue  # only frames that have detections are fed (motcpp_eval.cpp:314-319)
            want += mot_lines(trk.update(d), f)
        got = open(os.path.join(res, seq + ".txt")).read()
        assert got == "".join(want), seq


@pytest.mark.gpu
def test_cli_ucmc_matches_oracle_on_the_mot17_detections(tmp_path, orc):
    """UCMCTrack through the command-line tool with the reference tool's values (motcpp_eval.cpp:112-131: dt = 1.0 / fps, no camera
    file) on the real MOT17 detections of the fixture: the result files equal the oracle's tables line for line."""
    build()
    root, res = make_root(str(tmp_path)), os.path.join(str(tmp_path), "results")
    out = subprocess.run([EVAL, root, res, "ucmc"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = 0
    for seq in mot17.SEQS:
        trk = orc.ucmc([0.3, 30, 100.0, 100.0, 5.0, 5.0, 10.0, 1.0 / 25, 0.5])
        want = []
        for f, d in enumerate(mot17.load(seq), start=1):
            if d.shape[0] == 0:
                continue
            want += mot_lines(trk.update(d), f)
        assert open(os.path.join(res, seq + ".txt")).read() == "".join(want), seq
        rows += len(want)
    assert rows > 1000


@pytest.mark.gpu
def test_cli_boosttrack_matches_oracle_on_the_mot17_detections(tmp_path, orc):
    """BoostTrack++ (motion only) through the command-line tool with boosttrack.yaml's values (motcpp_eval.cpp:247-278: use_sb and use_vt on)
    on the real MOT17 detections of the fixture: result files equal to the oracle's tables line for line."""
    build()
    root, res = make_root(str(tmp_path)), os.path.join(str(tmp_path), "results")
    out = subprocess.run([EVAL, root, res, "boosttrack"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = 0
    for seq in mot17.SEQS:
        trk = orc.tracker(orclib.BOOSTTRACK, [0.6, 60, 3, 0.3, 10, 1.6, 0.5, 0.25, 0.25, 1, 1, 0.65, 1, 1])
        want = []
        for f, d in enumerate(mot17.load(seq), start=1):
            if d.shape[0] == 0:
                continue
            want += mot_lines(trk.update(d), f)
        assert open(os.path.join(res, seq + ".txt")).read() == "".join(want), seq
        rows += len(want)
    assert rows > 1000


@pytest.mark.gpu
def test_cli_hybridsort_matches_oracle_on_the_mot17_detections(tmp_path, orc):
    """HybridSORT through the command-line tool with hybridsort.yaml's values (motcpp_eval.cpp:279-316; no ReID weights: with_reid = false) on
    the real MOT17 detections of the fixture: result files equal to the oracle's tables line for line."""
    build()
    root, res = make_root(str(tmp_path)), os.path.join(str(tmp_path), "results")
    out = subprocess.run([EVAL, root, res, "hybridsort"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = 0
    for seq in mot17.SEQS:
        trk = orc.tracker(orclib.HYBRIDSORT, [0.5, 30, 3, 0.3, 1, 0.1, 1, 0.5, 4.6, 1.3, 1, 1, 1.0, 0])
        want = []
        for f, d in enumerate(mot17.load(seq), start=1):
            if d.shape[0] == 0:
                continue
            want += mot_lines(trk.update(d), f)
        assert open(os.path.join(res, seq + ".txt")).read() == "".join(want), seq
        rows += len(want)
    assert rows > 1000


@pytest.mark.gpu
def test_cli_ablation_offset(tmp_path, orc):
    build()
    gt = {"MOT17-02-FRCNN": 300, "MOT17-04-FRCNN": 1050}  # 02: det frames run to 600 > 1.5 x 300 -> offset 300; 04: no offset
    root, res = make_root(str(tmp_path), gt_frames=gt), os.path.join(str(tmp_path), "results")
    out = subprocess.run([EVAL, root, res, "sort"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "skipping the first 300 frames" in out.stdout
    trk = orc.tracker(orclib.SORT, [0.3, 1, 50, 3, 0.3])
    want = []
    for f, d in enumerate(mot17.load("MOT17-02-FRCNN"), start=1):
        if d.shape[0] == 0 or f <= 300:
            continue  # frames at or below the offset never reach the tracker
        want += mot_lines(trk.update(d), f - 300)
    assert open(os.path.join(res, "MOT17-02-FRCNN.txt")).read() == "".join(want)
    first = open(os.path.join(res, "MOT17-04-FRCNN.txt")).readline()
    assert int(first.split(",")[0]) >= 1 and "1050" in open(os.path.join(res, "MOT17-04-FRCNN.txt")).read()


def make_pregenerated(tmp, D=8, frames=12, seed=3):
    """<tmp>/pre/yolox/{dets,embs/osnet}/MOT17-02.txt: whitespace-separated detections (sorted by frame) + one feature line per detection,
    in the same order; returns (det_emb_root, per-frame dets, per-frame embs)."""
    r = np.random.default_rng(seed)
    base = os.path.join(tmp, "pre")
    os.makedirs(os.path.join(base, "yolox", "dets"))
    os.makedirs(os.path.join(base, "yolox", "embs", "osnet"))
    dets, embs = {}, {}
    proto = r.standard_normal((6, D)).astype(np.float32)
    with open(os.path.join(base, "yolox", "dets", "MOT17-02.txt"), "w") as fd, open(os.path.join(base, "yolox", "embs", "osnet", "MOT17-02.txt"), "w") as fe:
        fe.write("# one line per detection\n")
        for f in range(1, frames + 1):
            if f == 5:
                continue  # a frame without detections
            n = 6 if f % 3 else 4
            d = np.zeros((n, 6), np.float32)
            for i in range(n):
                x, y = 100 + 220 * i + 3 * f, 200 + 2 * f
                d[i] = [x, y, x + 60, y + 140, 0.9 - 0.05 * i, 0]
                fd.write("%d %.2f %.2f %.2f %.2f %.3f 0\n" % (f, *d[i, :5]))
            e = (proto[:n] + 0.05 * r.standard_normal((n, D))).astype(np.float32)
            for row in e:
                fe.write(" ".join("%.6f" % v for v in row) + "\n")
                fe.write("\n" if f == 2 else "")  # empty lines are skipped
            dets[f], embs[f] = np.round(d, 3), e
        fe.write(" ".join(["9"] * D) + "\n")  # one line more than there are detections: ignored
    return base, dets, embs


def test_embedding_file_reader(tmp_path):
    root = make_root(str(tmp_path))
    base, dets, embs = make_pregenerated(str(tmp_path))
    out = subprocess.run([build(), "embs", root, base, "yolox", "osnet"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    seqs = [l.split() for l in out.stdout.splitlines() if l.startswith("EMB ")]
    assert ["EMB", "MOT17-02-FRCNN", "file=MOT17-02.txt", "frames=%d" % len(embs)] in seqs
    rows = [dict(p.split("=") for p in l.split()[2:]) | {"f": l.split()[1]} for l in out.stdout.splitlines() if l.startswith("F ")]
    assert [int(r["f"]) for r in rows] == sorted(embs)
    for r in rows:
        e = np.loadtxt([" ".join("%.6f" % v for v in row) for row in embs[int(r["f"])]], dtype=np.float32, ndmin=2)
        assert int(r["rows"]) == int(r["dets"]) == e.shape[0] and int(r["d"]) == e.shape[1]
        want = float(np.sum(e.astype(np.float64) * np.arange(1, e.shape[0] + 1)[:, None] * np.arange(1, e.shape[1] + 1)[None, :]))
        assert abs(float(r["sum"]) - want) < 1e-3 * max(1.0, abs(want)), r


@pytest.mark.gpu
def test_cli_botsort_with_embedding_files(tmp_path, orc):
    """motcpp_eval <mot_root> <out> botsort <det_emb_root> <model> <reid>: pre-generated detections and one feature per detection reach
    BoT-SORT's update(dets, img, embs) — the result file against the oracle fed the same rows"""
    build()
    root = make_root(str(tmp_path))
    base, dets, embs = make_pregenerated(str(tmp_path), D=16, frames=30)
    res = os.path.join(str(tmp_path), "results")
    out = subprocess.run([EVAL, root, res, "botsort", base, "yolox", "osnet"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "embeddings:" in out.stdout and "(%d frames)" % len(embs) in out.stdout
    # the tool's BoT-SORT preset (tools/motcpp_eval.cpp) with with_reid on, no camera-motion compensation
    trk = orc.tracker(orclib.BOTSORT, [0.6, 0.1, 0.7, 30, 0.8, 0.5, 0.25, 25, 0, 1])
    want = []
    for f in sorted(dets):
        d = np.loadtxt(["%.2f %.2f %.2f %.2f %.3f 0" % tuple(r[:5]) for r in dets[f]], dtype=np.float32, ndmin=2)
        e = np.loadtxt([" ".join("%.6f" % v for v in row) for row in embs[f]], dtype=np.float32, ndmin=2)
        want += mot_lines(trk.update(d, e), f)
    assert open(os.path.join(res, "MOT17-02-FRCNN.txt")).read() == "".join(want)
