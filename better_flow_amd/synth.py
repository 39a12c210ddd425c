"""Deterministic synthetic event streams (SURVEY.md section 8(d) generator).

A scene of P = N/16 points uniformly placed on the sensor moves with a constant
image-plane velocity; every event picks a scene point and a timestamp.  The PRNG
is a counter-based splitmix64 written out here (no numpy/std distributions), so
the same stream can be regenerated bit-identically anywhere.

Convention (bf_motion_compensator.cpp:192,200 of the reference): ``fr_x`` is the
sensor ROW in [0, H), ``fr_y`` the COLUMN in [0, W).
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, stream, n):
    """n 64-bit outputs of splitmix64 for (seed, stream); counter based."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(stream) * np.uint64(
            0xA0761D6478BD642F
        )
        z = base + (np.arange(1, n + 1, dtype=np.uint64)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform(seed, stream, n):
    """Doubles in [0, 1) from the top 53 bits."""
    return (splitmix64(seed, stream, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def default_velocity(height, width):
    """(v_row, v_col) px/s of the SURVEY generator: (-150*H/180, +300*W/240)."""
    return (-150.0 * height / 180.0, 300.0 * width / 240.0)


def make_slice(n_events, height, width, duration_s=0.030, seed=1, velocity=None, t0_ns=0):
    """Return dict(fr_x, fr_y: int32; t: int64 ns from slice start, ascending).

    Events whose scene point has left the sensor are dropped, so the returned
    count is <= n_events (a few % fewer)."""
    if velocity is None:
        velocity = default_velocity(height, width)
    v_row, v_col = velocity
    n_pts = max(1, n_events // 16)
    p_row = 10.0 + _uniform(seed, 1, n_pts) * (height - 20.0)
    p_col = 10.0 + _uniform(seed, 2, n_pts) * (width - 20.0)
    t_ns = np.floor(_uniform(seed, 3, n_events) * (duration_s * 1e9)).astype(np.int64)
    t_ns.sort(kind="stable")
    pick = (splitmix64(seed, 4, n_events) % np.uint64(n_pts)).astype(np.int64)
    ts = t_ns.astype(np.float64) * 1e-9
    row = np.floor(p_row[pick] + v_row * ts).astype(np.int64)
    col = np.floor(p_col[pick] + v_col * ts).astype(np.int64)
    keep = (row >= 0) & (row < height) & (col >= 0) & (col < width)
    return {
        "fr_x": row[keep].astype(np.int32),
        "fr_y": col[keep].astype(np.int32),
        "t": (t_ns[keep] + np.int64(t0_ns)).astype(np.int64),
        "height": height,
        "width": width,
        "velocity": (v_row, v_col),
    }


def write_txt(path, sl, time_offset_s=1.0):
    """Text format the reference CLI reads: ``t x y p`` = seconds, COLUMN, ROW, polarity
    (bf_motion_compensator.cpp:190-202)."""
    with open(path, "w") as f:
        for t, r, c in zip(sl["t"], sl["fr_x"], sl["fr_y"]):
            f.write("%.9f %d %d 1\n" % (time_offset_s + t * 1e-9, c, r))


def write_bin(path, sl, time_offset_s=1.0):
    """Binary structure-of-arrays event file (better_flow_amd/host/better_flow/event_reader.h): magic
    "BFEVSOA1", u64 n, u64 t_ns[n], u16 x[n] (column), u16 y[n] (row), u8 p[n].  Times are what the CLI
    derives from the text form written by write_txt: ull(1e9 * (t - t_0)) on the printed decimals."""
    import numpy as np
    txt = np.array([float("%.9f" % (time_offset_s + t * 1e-9)) for t in sl["t"]])
    rel = txt - txt[0] if len(txt) else txt
    t_ns = (1000000000 * rel).astype(np.uint64)      # truncation, like the reference's FROM_SEC
    with open(path, "wb") as f:
        f.write(b"BFEVSOA1")
        f.write(np.uint64(len(t_ns)).tobytes())
        f.write(t_ns.astype("<u8").tobytes())
        f.write(np.asarray(sl["fr_y"]).astype("<u2").tobytes())
        f.write(np.asarray(sl["fr_x"]).astype("<u2").tobytes())
        f.write(np.ones(len(t_ns), dtype=np.uint8).tobytes())


def write_stream_bin(path, n_slices, events_per_slice, height, width, duration_s=0.030, seed=1, distinct=4):
    """A rolling stream for the front-end measurements: n_slices consecutive `duration_s` blocks of ~events_per_slice
    events, as one binary structure-of-arrays event file.  Only `distinct` different blocks are generated (seeds seed,
    seed + 1, ...) and repeated round-robin with their times shifted, so a 20M-event file takes seconds to write.
    Returns the number of events."""
    blocks = [make_slice(events_per_slice, height, width, duration_s, seed=seed + k) for k in range(min(distinct, n_slices))]
    step = np.uint64(int(round(duration_s * 1e9)))
    with open(path, "wb") as f:
        total = sum(len(blocks[i % len(blocks)]["t"]) for i in range(n_slices))
        f.write(b"BFEVSOA1")
        f.write(np.uint64(total).tobytes())
        for i in range(n_slices):
            f.write((blocks[i % len(blocks)]["t"].astype(np.uint64) + np.uint64(i) * step + np.uint64(1_000_000_000)).astype("<u8").tobytes())
        for i in range(n_slices):
            f.write(blocks[i % len(blocks)]["fr_y"].astype("<u2").tobytes())
        for i in range(n_slices):
            f.write(blocks[i % len(blocks)]["fr_x"].astype("<u2").tobytes())
        f.write(np.ones(total, dtype=np.uint8).tobytes())
    return total
