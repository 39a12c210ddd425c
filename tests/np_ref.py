"""Independent numpy restatement of the reference arithmetic for
SMALL cases.  Used only to cross-check the C oracle: two restatements written from the
same reference lines must agree bit for bit.  Test infrastructure only."""
import numpy as np

F32 = np.float32


def window(fr_x, fr_y, scale, res_x, res_y):
    """optimizer_rolling.h:248-283."""
    x_min = min([res_x] + [int(v) for v in fr_x])
    y_min = min([res_y] + [int(v) for v in fr_y])
    x_max = max([0] + [int(v) for v in fr_x])
    y_max = max([0] + [int(v) for v in fr_y])
    wsx, wsy = scale * (x_max - x_min), scale * (y_max - y_min)
    xs = -float((x_max - x_min) // 2 + x_min) * float(scale) + float(wsx) / 2.0 + scale // 2
    ys = -float((y_max - y_min) // 2 + y_min) * float(scale) + float(wsy) / 2.0 + scale // 2
    return dict(x_min=x_min, y_min=y_min, x_max=x_max, y_max=y_max, wsx=wsx, wsy=wsy,
                R=wsx + scale, C=wsy + scale, x_shift=xs, y_shift=ys)


def warp(fr_x, fr_y, t, pr_x, pr_y, dnx, dny, cx, cy, div, crl):
    """event.h:99-110,164-168 (vectorised; every op is a separate IEEE operation)."""
    rx, ry = pr_x - cx, pr_y - cy
    c, s = np.cos(crl), np.sin(crl)
    qx = c * rx - s * ry
    qy = s * rx + c * ry
    nx = ((-qx) * div + (qx - rx)) + dnx
    ny = ((-qy) * div + (qy - ry)) + dny
    kx = (nx.astype(F32).astype(np.float64) / 127.0).astype(F32)
    ky = (ny.astype(F32).astype(np.float64) / 127.0).astype(F32)
    ft = t.astype(F32)
    px = (kx * ft).astype(np.float64)           # f32 product
    py = (ky * ft).astype(np.float64)
    npr_x = fr_x.astype(F32).astype(np.float64) - px / 10000.0
    npr_y = fr_y.astype(F32).astype(np.float64) - py / 10000.0
    return npr_x, npr_y, nx, ny


def time_img(pr_x, pr_y, t, w, scale, noise=None):
    """accel_lib.h:147-178, event by event in container order."""
    R, C = w["R"], w["C"]
    avg = np.zeros((R, C), F32)
    cnt = np.zeros((R, C), F32)
    x_sh, y_sh = int(w["x_shift"]), int(w["y_shift"])
    hs = scale // 2
    for i in range(len(t)):
        if noise is not None and noise[i]:
            continue
        x = int(np.trunc(pr_x[i] * scale + x_sh))
        y = int(np.trunc(pr_y[i] * scale + y_sh))
        if x >= w["wsx"] + hs or x < hs or y >= w["wsy"] + hs or y < hs:
            continue
        for jx in range(x - hs, x + hs + 1):
            for jy in range(y - hs, y + hs + 1):
                avg[jx, jy] = F32(np.float64(avg[jx, jy]) + np.float64(t[i]) / 1000000000.0)
                cnt[jx, jy] = cnt[jx, jy] + F32(1)
    m = cnt >= 1
    avg[m] = avg[m] / cnt[m]
    return avg, cnt


def scharr(img):
    """accel_lib.h:513-615."""
    R, C = img.shape
    gx = np.zeros((R, C), F32)
    gy = np.zeros((R, C), F32)
    sx = [3, 0, -3, 10, 0, -10, 3, 0, -3]
    sy = [3, 10, 3, 0, 0, 0, -3, -10, -3]
    for i in range(1, R - 1):
        for j in range(1, C - 1):
            if not (np.float64(img[i, j]) > 0.000001):
                continue
            dx = F32(0)
            dy = F32(0)
            ok = True
            idx = 0
            for k in range(3):
                for l in range(3):
                    val = img[l + i - 1, k + j - 1]
                    if np.float64(val) <= 0.000001:
                        ok = False
                        break
                    dx = F32(dx + F32(val * F32(sx[idx])))
                    dy = F32(dy + F32(val * F32(sy[idx])))
                    idx += 1
                if not ok:
                    break
            if ok:
                gx[i, j], gy[i, j] = dx, dy
    return gx, gy


def model(img):
    """object_model.cpp:4-39,103-126 (row-major double accumulation)."""
    gx, gy = scharr(img)
    R, C = img.shape
    valid = img.astype(np.float64) > 0.000001
    ii, jj = np.nonzero(valid)
    n = len(ii)
    cx = float(ii.sum()) / n
    cy = float(jj.sum()) / n
    dx = dy = rot = div = 0.0
    for i, j in zip(ii, jj):
        rx, ry = float(i) - cx, float(j) - cy
        g0, g1 = float(gx[i, j]), float(gy[i, j])
        dx += g0
        dy += g1
        rot += rx * g1 - ry * g0
        div += rx * g0 + ry * g1
    return dict(cx=cx, cy=cy, dx=dx / n, dy=dy / n, rot=rot / n, div=div / n, cnt=n)
