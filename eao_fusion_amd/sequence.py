"""The batched-sequence configuration (BASELINE configs[4]): one rank's share of a sequence of
frames -- ORB extraction of its contiguous frame shard, brute-force best-2 matching of consecutive frames (pair (f - 1, f)
belongs to the shard of f: the first pair needs the LAST frame of the previous shard, the halo), and its share of the
independent local-BA windows (window w -> rank w mod N).  Everything between the frames arriving in HBM and the results is
device-resident: the matcher reads the extractor's descriptor block and keypoint counts in place.

torch is plumbing here (device buffers, the stream handle); every computation is a libeaofusion_hip.so entry point."""
import ctypes as C

import numpy as np
import torch

from . import _lib, shard, synth
from .optimizer import Optimizer
from .orb import ORBextractor

N_WINDOWS = 25          # BA windows of the 512-frame sequence, generated as config 4 with seed 6000 + w


def sequence_frame(f, w=640, h=480):
    """Frame f of the synthetic sequence (seed 5000 + f)."""
    return synth.synth_frame(5000 + f, w, h)


def window_problem(w):
    return synth.synth_ba(seed=6000 + w)


class SequenceShard:
    """Device buffers and the three enqueues of one shard of `batch` frames."""

    def __init__(self, batch, width=640, height=480, device=None, nfeatures=1000):
        self.dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.B, self.W, self.H = batch, width, height
        self.ext = ORBextractor(nfeatures, 1.2, 8, 20, 7)
        self.cap = self.ext.max_keypoints(width, height)
        self.d_kps = torch.zeros((batch, self.cap, 28), dtype=torch.uint8, device=self.dev)
        self.d_desc = torch.zeros((batch, self.cap, 32), dtype=torch.uint8, device=self.dev)
        self.d_n = torch.zeros(batch, dtype=torch.int32, device=self.dev)
        self.d_match = torch.zeros((batch, self.cap, 4), dtype=torch.int32, device=self.dev)    # eao_best2 records
        self._L = _lib.load()

    def extract(self, d_img):
        """d_img: [B, H, W] uint8 on the device.  Asynchronous on torch's current stream."""
        st = torch.cuda.current_stream().cuda_stream
        self.ext.extract_batch_device(d_img.data_ptr(), self.W, self.H, self.W, self.W * self.H, self.B, self.d_kps.data_ptr(),
                                      self.d_desc.data_ptr(), self.cap, self.d_n.data_ptr(), st)

    def match(self, halo_desc=None, halo_n=0):
        """Pairs (f - 1, f) of the shard; pair 0 against the halo frame's descriptors [cap, 32] (or skipped).  Rows that no pair
        writes stay -1."""
        st = torch.cuda.current_stream().cuda_stream
        self.d_match.fill_(-1)
        hp = halo_desc.data_ptr() if (halo_desc is not None and halo_n) else None
        _lib.check(self._L.eao_hamming_best2_sequence_device(self.d_desc.data_ptr(), self.cap, self.d_n.data_ptr(), self.B, hp,
                                                             int(halo_n) if hp else 0, self.d_match.data_ptr(), st))

    def last_frame(self):
        """(keypoints, descriptors, count) of the shard's last frame: the halo of the next shard."""
        return self.d_kps[self.B - 1], self.d_desc[self.B - 1], int(self.d_n[self.B - 1].item())


def run_windows(windows, packed=None):
    """This rank's local-BA windows through ONE eao_local_ba_batch call.  Returns (cams [n, n_cams, 16], points [n, n_points, 3],
    residual blocks linearised, list of result dicts)."""
    if not windows:
        return np.zeros((0, 24, 16), np.float32), np.zeros((0, 3000, 3), np.float32), 0, []
    probs = [window_problem(w) for w in windows] if packed is None else None
    res = Optimizer.LocalBundleAdjustmentBatch(probs, packed=packed)
    cams = np.stack([r["poses"].reshape(-1, 16) for r in res])
    pts = np.stack([r["points"] for r in res])
    return cams, pts, res[0]["timing"]["linearizations"], res


def run_shard(frames, lo, rank, world, halo=None, n_windows=N_WINDOWS, device=None, seq=None):
    """Everything rank `rank` of `world` does for the sequence: frames = its [B, H, W] uint8 shard (global frames lo ..),
    halo = (desc [cap, 32] device tensor, count) of frame lo - 1 or None.  Returns a dict of device tensors / arrays."""
    B, H, W = frames.shape
    s = seq or SequenceShard(B, W, H, device)
    d_img = torch.from_numpy(frames).to(s.dev) if isinstance(frames, np.ndarray) else frames
    s.extract(d_img)
    s.match(*(halo if halo else (None, 0)))
    torch.cuda.synchronize()
    cams, pts, lin, res = run_windows(shard.window_shard(n_windows, rank, world))
    return dict(seq=s, kps=s.d_kps, desc=s.d_desc, n=s.d_n, match=s.d_match, ba_cams=cams, ba_points=pts, ba_linearizations=lin, ba_results=res)
