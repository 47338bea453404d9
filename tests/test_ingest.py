"""frame ingest.  CPU part (no GPU needed): directory listing / timestamp stems as
/root/reference/src/ImageReader.cpp:22-78 does them, PGM (P5), raw and greyscale PNG decode in place of cv::imread.
GPU part: the pinned double-buffered feeder feeds the pipeline through files -> reader -> pinned buffer -> device and
the result is identical to running on frames that were resident all along."""
import os

import numpy as np
import pytest


def _write_pgm(path, img, comment=None, maxval=255):
    with open(path, "wb") as f:
        f.write(b"P5\n")
        if comment:
            f.write(b"# " + comment.encode() + b"\n")
        f.write(f"{img.shape[1]} {img.shape[0]}\n{maxval}\n".encode())
        f.write(img.tobytes())


def test_listing_order_and_timestamps(vislam, tmp_path):
    rng = np.random.default_rng(3)
    names = ["1403636579763555584.pgm", "1403636579813555456.pgm", "1403636579713555456.pgm", "0000000000000000100.raw"]
    for n in names:
        (tmp_path / n).write_bytes(rng.integers(0, 256, 64, dtype=np.uint8).tobytes())
    (tmp_path / "notes.txt").write_text("not an image")
    (tmp_path / "sub.pgm").mkdir()                       # a directory that looks like an image is not listed
    got = vislam.image_list(str(tmp_path))
    assert got == sorted(names)                          # byte order, as std::sort on the names
    # EuRoC stems are nanosecond timestamps: getImageTime = atol(stem); computeTimeStep = difference of the first two
    assert vislam.image_time(got[1]) == 1403636579713555456
    assert vislam.image_time("/data/cam0/" + got[2]) == 1403636579763555584
    assert vislam.image_time(got[2]) - vislam.image_time(got[1]) == 50000128
    assert vislam.image_time("frame.pgm") == 0           # atol of a non-numeric stem
    with pytest.raises(vislam.VisError):
        vislam.image_list(str(tmp_path / "missing"))


def test_pgm_and_raw_decode(vislam, tmp_path):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (48, 80), dtype=np.uint8)
    p = str(tmp_path / "a.pgm")
    _write_pgm(p, img, comment="EuRoC cam0 frame")
    assert (vislam.image_read(p) == img).all()
    # whitespace variants of the header and a pixel byte that looks like whitespace right after it
    img2 = img.copy(); img2[0, 0] = 10                   # '\n' as first pixel must not be eaten by the header parser
    with open(str(tmp_path / "b.pgm"), "wb") as f:
        f.write(b"P5 80\t48 # size\n255\n" + img2.tobytes())
    assert (vislam.image_read(str(tmp_path / "b.pgm")) == img2).all()
    # raw: headerless
    (tmp_path / "c.raw").write_bytes(img.tobytes())
    assert (vislam.image_read(str(tmp_path / "c.raw"), 80, 48) == img).all()
    # errors: truncated payload, 16-bit PGM, ASCII PGM, wrong size, missing file
    with open(str(tmp_path / "t.pgm"), "wb") as f:
        f.write(b"P5\n80 48\n255\n" + img.tobytes()[:-5])
    with pytest.raises(vislam.VisError):
        vislam.image_read(str(tmp_path / "t.pgm"))
    _write_pgm(str(tmp_path / "w.pgm"), img, maxval=65535)
    with pytest.raises(vislam.VisError):
        vislam.image_read(str(tmp_path / "w.pgm"))
    (tmp_path / "p2.pgm").write_bytes(b"P2\n2 2\n255\n1 2 3 4\n")
    with pytest.raises(vislam.VisError):
        vislam.image_read(str(tmp_path / "p2.pgm"))
    with pytest.raises(vislam.VisError):
        vislam.image_read(p, 64, 48)
    with pytest.raises(vislam.VisError):
        vislam.image_read(str(tmp_path / "nope.pgm"))


def _png_chunk(tag, data):
    import struct
    import zlib
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _write_png(path, samples, depth=8, color=0, filters=None, idat_split=None, interlace=0, corrupt_crc=False, level=6):
    """a PNG writer for the tests, from the specification: `samples` = (h, w, channels) integer array of `depth`-bit samples;
    `filters` = the filter type of every row (default: 0..4 in turn); the scanline filters are applied here, byte by byte"""
    import struct
    import zlib
    h, w, ch = samples.shape
    bps = depth // 8
    if depth == 16:
        rows = samples.astype(">u2").tobytes()
    else:
        rows = samples.astype(np.uint8).tobytes()
    rowb, bpp = w * ch * bps, ch * bps
    raw = bytearray()
    prev = bytes(rowb)
    for y in range(h):
        cur = rows[y * rowb:(y + 1) * rowb]
        ft = (filters[y] if filters is not None else y % 5)
        if ft in (0, 2):                                             # vectorised (whole frames in the directory test)
            ca, pa_ = np.frombuffer(cur, np.uint8), np.frombuffer(prev, np.uint8)
            raw += bytes([ft]) + (ca if ft == 0 else (ca - pa_)).astype(np.uint8).tobytes()
            prev = cur
            continue
        out = bytearray(rowb)
        for i in range(rowb):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pr = 0
            elif ft == 1:
                pr = a
            elif ft == 2:
                pr = b
            elif ft == 3:
                pr = (a + b) >> 1
            else:
                pp = a + b - c
                pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (cur[i] - pr) & 0xFF
        raw += bytes([ft]) + out
        prev = cur
    z = zlib.compress(bytes(raw), level)
    parts = [z] if not idat_split else [z[i:i + idat_split] for i in range(0, len(z), idat_split)]
    blob = b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, interlace))
    blob += _png_chunk(b"tEXt", b"Comment\x00EuRoC-shaped test frame")           # an ancillary chunk in front of the data
    for part in parts:
        blob += _png_chunk(b"IDAT", part)
    blob += _png_chunk(b"IEND", b"")
    if corrupt_crc:
        i = blob.index(b"IDAT") + 8
        blob = blob[:i] + bytes([blob[i] ^ 0x40]) + blob[i + 1:]
    with open(path, "wb") as f:
        f.write(blob)


def test_png_decode_greyscale_as_imread_would(vislam, tmp_path):
    """EuRoC ships cam0 as 8-bit greyscale PNG (the reference decodes it with imread(..., CV_LOAD_IMAGE_GRAYSCALE), src/ImageReader.cpp:80-82).
    Every scanline filter, several IDAT chunks, 16-bit samples (high byte), grey + alpha (alpha dropped), a 752x480 frame; what imread would
    push through libpng's colour conversion (RGB, palette) or de-interlacing is refused, as are damaged files."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    img[5:9, :] = 0
    img[20, :] = 255                                                   # rows that wrap around in the filters' byte arithmetic
    p = str(tmp_path / "1403636579763555584.png")
    _write_png(p, img[:, :, None])
    assert (vislam.image_read(p) == img).all()
    for ft in range(5):                                                # one filter type for the whole image, each in turn
        _write_png(p, img[:, :, None], filters=[ft] * img.shape[0], idat_split=97)
        assert (vislam.image_read(p) == img).all(), ft
    g16 = rng.integers(0, 65536, (19, 31), dtype=np.uint16)
    _write_png(p, g16[:, :, None], depth=16)
    assert (vislam.image_read(p) == (g16 >> 8).astype(np.uint8)).all()
    ga = rng.integers(0, 256, (23, 29, 2), dtype=np.uint8)
    _write_png(p, ga, color=4)
    assert (vislam.image_read(p) == ga[:, :, 0]).all()
    ga16 = rng.integers(0, 65536, (9, 14, 2), dtype=np.uint16)
    _write_png(p, ga16, depth=16, color=4)
    assert (vislam.image_read(p) == (ga16[:, :, 0] >> 8).astype(np.uint8)).all()
    frame = vislam.synth_frame(vislam.synth_canvas(1024, 5), 3, 752, 480, 5)
    _write_png(p, frame[:, :, None], level=1)
    assert (vislam.image_read(p) == frame).all() and (vislam.image_read(p, 752, 480) == frame).all()
    assert vislam.image_list(str(tmp_path)) == ["1403636579763555584.png"] and vislam.image_time(p) == 1403636579763555584
    # refused, not guessed
    rgb = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    bad = str(tmp_path / "bad.png")
    for kw, arr in ((dict(color=2), rgb), (dict(interlace=1), img[:, :, None]), (dict(corrupt_crc=True), img[:, :, None])):
        _write_png(bad, arr, **kw)
        with pytest.raises(vislam.VisError):
            vislam.image_read(bad)
    _write_png(bad, img[:, :, None])
    blob = open(bad, "rb").read()
    open(bad, "wb").write(blob[:len(blob) // 2])                       # truncated
    with pytest.raises(vislam.VisError):
        vislam.image_read(bad)
    _write_png(bad, img[:, :, None])
    with pytest.raises(vislam.VisError):
        vislam.image_read(bad, 64, 48)                                 # wrong size requested


def test_png_reader_survives_damaged_files(vislam, tmp_path):
    """image files are untrusted input: 240 mutations of a valid PNG (byte flips, truncations, chunk-length / size fields rewritten with
    their CRC made valid again) either decode to an image of the requested size or return an error -- no crash, no exception across the C
    boundary, no read past a buffer (the test-suite's sanitizer run covers the oracle only; this is the product's host code)."""
    import struct
    import zlib
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (24, 40), dtype=np.uint8)
    p = str(tmp_path / "ok.png")
    _write_png(p, img[:, :, None], idat_split=50)
    good = open(p, "rb").read()
    assert (vislam.image_read(p) == img).all()
    bad = str(tmp_path / "m.png")
    outcomes = {"decoded": 0, "refused": 0}
    for i in range(240):
        b = bytearray(good)
        kind = i % 4
        if kind == 0:                                                  # random byte flips anywhere
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(8, len(b)))] ^= int(rng.integers(1, 256))
        elif kind == 1:                                                # truncation
            b = b[:int(rng.integers(0, len(b)))]
        elif kind == 2:                                                # IHDR fields rewritten, CRC valid
            w_, h_ = [int(x) for x in rng.choice([0, 1, 40, 24, 41, 4096, 70000, 2 ** 31 - 1], 2)]
            ihdr = struct.pack(">IIBBBBB", w_, h_, int(rng.choice([1, 8, 16])), int(rng.choice([0, 2, 3, 4, 6])), 0, 0, int(rng.choice([0, 1])))
            b[8:33] = _png_chunk(b"IHDR", ihdr)
        else:                                                          # a chunk length field rewritten
            pos = 8
            offs = []
            while pos + 12 <= len(b):
                offs.append(pos)
                pos += 12 + struct.unpack(">I", bytes(b[pos:pos + 4]))[0]
            o = int(rng.choice(offs))
            b[o:o + 4] = struct.pack(">I", int(rng.choice([0, 1, 13, 2 ** 31, 2 ** 32 - 1, len(b)])))
        open(bad, "wb").write(bytes(b))
        for args in ((), (40, 24)):
            try:
                out = vislam.image_read(bad, *args)
                assert out.dtype == np.uint8 and out.ndim == 2
                outcomes["decoded"] += 1
            except vislam.VisError:
                outcomes["refused"] += 1
            except MemoryError:                                        # the WRAPPER allocating what a forged header claims: refused before the library is asked
                outcomes["refused"] += 1
    assert outcomes["refused"] > 360 and outcomes["decoded"] >= 0, outcomes


@pytest.mark.gpu
def test_feeder_round_trip_and_pipeline_equality(vislam, ctx, canvas, tmp_path):
    import torch
    W, H, B = 752, 480, 6
    frames = np.stack([vislam.synth_frame(canvas, t, W, H) for t in range(2 * B)])
    # frames go through files -> reader -> pinned buffer -> device
    for t in range(2 * B):
        _write_pgm(str(tmp_path / f"{1403636579763555584 + 50000000 * t}.pgm"), frames[t])
    names = vislam.image_list(str(tmp_path))
    assert len(names) == 2 * B
    p = vislam.default_params(); p.fy = p.fx
    ctx.set_params(p)
    ctx.batch_plan(W, H, W, B)
    feed = vislam.Feeder(ctx, W, H, B)
    got = []
    for k in range(2):
        hb = feed.host_buffer(k)
        for i in range(B):
            hb[i] = vislam.image_read(os.path.join(str(tmp_path), names[k * B + i]))
        d = feed.submit(k, B)
        ctx.batch_run(d, B)
        feed.release(k)
        ctx.batch_sync()
        got.append([ctx.batch_keypoints(i) for i in range(B)] + [ctx.batch_pose(i) for i in range(B)])
    feed.close()
    # same stream with resident frames (fresh plan state)
    ctx.batch_reset()
    d_all = torch.from_numpy(frames).cuda()
    for k in range(2):
        ctx.batch_run(d_all.data_ptr() + k * B * H * W, B)
        ctx.batch_sync()
        for i in range(B):
            k1, d1 = ctx.batch_keypoints(i)
            k0, d0 = got[k][i]
            assert (k1 == k0).all() and (d1 == d0).all(), (k, i)
            assert str(ctx.batch_pose(i)) == str(got[k][B + i]), (k, i)


@pytest.mark.gpu
def test_directory_harness_on_a_euroc_shaped_png_directory(built, vislam, canvas, tmp_path):
    """tools/run_directory.py = BASELINE configs[0] on a supplied dataset (none ships with the image): an EuRoC-shaped directory -- 752x480
    8-bit greyscale PNGs named by nanosecond timestamps 50 ms apart, plus a file that is not an image -- through listing, PNG decode, the
    pinned feeder and the batched pipeline with ORB::create(200); every frame checked against the CPU oracle by the tool itself."""
    import json
    import subprocess
    import sys
    d = tmp_path / "mav0" / "cam0" / "data"
    d.mkdir(parents=True)
    n = 12
    for t in range(n):
        _write_png(str(d / f"{1403636579763555584 + 50000000 * t}.png"), vislam.synth_frame(canvas, t, 752, 480)[:, :, None], filters=[0, 2] * 240, level=1)
    (d / "data.csv").write_text("#timestamp [ns],filename\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_directory.py"), str(d), "--frames", "200", "--batch", "5", "--check", str(n), "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["frames"] == n and j["width"] == 752 and j["height"] == 480 and j["first_timestamp"] == 1403636579763555584
    assert j["median_frame_interval_ns"] == 50000000 and j["checked_frames"] == n and j["frames_differing_from_the_oracle"] == []
    assert 180 <= j["keypoints_mean"] <= 260 and j["good_matches_mean"] > 10 and j["cpu_baseline"]["value"] > 0
