"""numpy restatement of the device RNG stream used by the reset kernel (csrc/gq_step_body.h): Philox4x32-10,
key = (seed_lo, seed_hi), counter = (draw >> 2, episode, env, 0x5eed), uniform = (word >> 8) * 2^-24.
Draw indices: 0-11 joint angle noise, 12-23 joint velocity noise, 24 x, 25 y, 26 roll, 27 pitch, 28 |v| command,
29 heading, 30 yaw rate, 31 friction, 32 command redraw interval (1000 + floor(2000 u); 'reset' command types).
In-episode redraws (gq_batch_set_resampling): command = block (0, n, env, 0xc0de): words |v|, heading, yaw rate, interval;
disturbance wrench = blocks (0..1, n, env, 0xd157): words 0-5 x y z roll pitch yaw, word 6 interval; n = redraws so far."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32(counter, key):
    c = [int(x) & 0xffffffff for x in counter]
    k = [int(x) & 0xffffffff for x in key]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
        k = [(k[0] + W0) & 0xffffffff, (k[1] + W1) & 0xffffffff]
    return c


def draws(seed, env, episode):
    """The 36 float32 uniforms of one reset."""
    u = np.zeros(36, dtype=np.float32)
    for blk in range(9):
        w = philox4x32((blk, episode, env, 0x5eed), (seed & 0xffffffff, seed >> 32))
        for j in range(4):
            u[4 * blk + j] = np.float32(w[j] >> 8) * np.float32(1.0 / 16777216.0)
    return u


def imu_normals(seed, env, step_num, episode=0):
    """The 12 standard normals of one IMU update (csrc gq_step_body.h): block counter (draw, step_num, env,
    0x1a70 ^ (episode << 8)), Box-Muller on words 0,1: acc noise xyz, acc bias step xyz, gyro noise xyz, gyro bias step xyz."""
    z = np.zeros(12, dtype=np.float64)
    for d in range(12):
        w = philox4x32((d, step_num, env, 0x1a70 ^ ((episode << 8) & 0xffffffff)), (seed & 0xffffffff, seed >> 32))
        u1 = (np.float32(w[0] >> 8) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        u2 = np.float32(w[1] >> 8) * np.float32(1.0 / 16777216.0)
        z[d] = np.sqrt(-2.0 * np.log(np.float64(u1))) * np.cos(2 * np.pi * np.float64(u2))
    return z


def resample_draws(seed, env, n, what):
    """The uniforms of the n-th in-episode redraw of env: what='cmd' -> 4 values, what='dist' -> 8 values."""
    tag, nblk = (0xc0de, 1) if what == 'cmd' else (0xd157, 2)
    u = np.zeros(4 * nblk, dtype=np.float32)
    for blk in range(nblk):
        w = philox4x32((blk, n, env, tag), (seed & 0xffffffff, seed >> 32))
        for j in range(4):
            u[4 * blk + j] = np.float32(w[j] >> 8) * np.float32(1.0 / 16777216.0)
    return u
