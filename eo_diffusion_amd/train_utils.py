"""Host-side pieces of the reference's training loop that are not kernels.

KeyframeLR: the keyframe learning-rate scheduler train.py builds at :76-85 (reference: script_utils/train_utils.py:17-226 -- that
module imports pytorch_lightning and timm at the top, so the reference's training script cannot even be imported without them; this
one needs torch only).  Same constructor, same positions / transitions vocabulary, same values (pinned by a reference-generated
fixture, tests/test_oracle_golden.py); the implementation is a sorted keyframe table with one bisect per step.

    scheduler = KeyframeLR(optimizer, units="steps", end=max_steps, frames=[
        {"position": 0, "lr": lr / 100}, {"transition": "cos"}, {"position": posmax, "lr": lr},
        {"transition": lambda last_lr, sf, ef, pos, *_: lr * math.exp(-3 * (pos - posmax) / (max_steps - posmax))}])
"""
import bisect
import math
import timeit
from collections.abc import Callable, Mapping, Sequence

from torch.optim.lr_scheduler import _LRScheduler

__all__ = ["KeyframeLR"]


def _ease(kind, t):
    """interpolation weight of the end keyframe at fraction t of the segment"""
    if kind == "linear":
        return t
    if kind == "cos":
        return 1.0 - (1.0 + math.cos(t * math.pi)) / 2.0
    raise ValueError(f"Unknown transition: {kind!r}")


class KeyframeLR(_LRScheduler):
    """Learning rate defined by keyframes {"position": p, "lr": v} (or the pair (p, v)) with an optional transition between two of
    them: "linear" (default), "cos", or a callable f(last_lr, start_frame, end_frame, position, scheduler).  `units`: "steps"
    (positions and `end` count .step() calls), "percent" (positions in [0, 1] of `end` steps) or "time" (fraction of `end` seconds).
    A schedule that does not start at position 0 ramps linearly from lr 0; one that stops before `end` ramps linearly to lr 0; a
    leading / trailing transition gets an implicit (0, 0) / (end, 0) keyframe.  "position": "end" is the last position."""

    def __init__(self, optimizer, frames, end, units="percent"):
        self.end = end
        self.units = units
        self.frames = self.parse_frames(frames)
        self.last_lr = 0
        self.start_time = timeit.default_timer() if units == "time" else None
        super().__init__(optimizer=optimizer)

    # ------------------------------------------------------------------ frame list
    def parse_frames(self, user_frames):
        last = self.end if self.units == "steps" else 1
        items = []
        for f in user_frames:
            if isinstance(f, (str, Callable)) and not isinstance(f, Mapping):
                f = {"transition": f}
            elif isinstance(f, Sequence) and not isinstance(f, (str, Mapping)) and len(f) == 2:
                f = {"position": f[0], "lr": f[1]}
            if f.get("position", None) == "end":
                f["position"] = last
            items.append(f)
        if not items:
            return []
        out = []
        # implicit start
        head = items[0]
        if "position" in head and head["position"] != 0:
            out += [{"position": 0, "lr": 0}, {"transition": "linear"}]
        if "transition" in head:
            out.append({"position": 0, "lr": 0})
        prev = -1
        for k, f in enumerate(items):
            out.append(f)
            if "position" in f:
                pos = f["position"]
                assert pos >= prev, f"position {pos!r} is not bigger than {prev}"
                assert pos <= last, f"position {pos} is bigger than end value {last}"
                prev = pos
                if k + 1 < len(items) and "position" in items[k + 1]:
                    out.append({"transition": "linear"})  # two keyframes in a row
        tail = items[-1]
        if "position" in tail and tail["position"] < last:
            out += [{"transition": "linear"}, {"position": last, "lr": 0}]
        if "transition" in tail:
            out.append({"position": last, "lr": 0})
        self._index(out)
        return out

    def _index(self, frames):
        """keyframes in order, and for each the transition that leads to the NEXT one (of several in a row the last counts)"""
        self._keys, self._kpos, self._trans = [], [], []
        pending = None
        for f in frames:
            if "position" in f:
                if self._keys:
                    self._trans.append(pending)
                self._keys.append(f)
                self._kpos.append(f["position"])
                pending = None
            elif self._keys:
                pending = f["transition"]

    @staticmethod
    def interpolate(a, b, pct):
        return (1 - pct) * a + pct * b

    def interpolate_frames(self, start_frame, transition, end_frame, position):
        if isinstance(transition, Callable) and not isinstance(transition, str):
            return transition(self.last_lr, start_frame, end_frame, position, self)
        span = end_frame["position"] - start_frame["position"]
        w = _ease(transition, (position - start_frame["position"]) / span)
        return self.interpolate(start_frame["lr"], end_frame["lr"], w)

    def get_lr_at_pos(self, position):
        kp = self._kpos
        i = bisect.bisect_left(kp, position)          # first keyframe at or behind `position`
        if i < len(kp) and kp[i] == position:
            lr = self._keys[i]["lr"]                  # on a keyframe (the first of several at one position)
        elif i == 0 or i == len(kp) or self._trans[i - 1] is None:
            print(f"No matching frames at position {position}, using last LR.")
            return self.last_lr
        else:
            lr = self.interpolate_frames(self._keys[i - 1], self._trans[i - 1], self._keys[i], position)
        self.last_lr = lr  # (kept for custom transitions and sample_lrs)
        return lr

    # ------------------------------------------------------------------ scheduler protocol
    @property
    def progress(self):
        if self.units == "time":
            return (timeit.default_timer() - self.start_time) / self.end
        return self.last_epoch / self.end

    def get_lr(self):
        if self.units == "percent":
            position = self.last_epoch / self.end
        elif self.units == "steps":
            position = self.last_epoch
        elif self.units == "time":
            position = (timeit.default_timer() - self.start_time) / self.end
        else:
            raise TypeError(f"Unknown units {self.units}")
        lr = self.get_lr_at_pos(position)
        return [lr for _ in self.optimizer.param_groups]

    def sample_lrs(self, n=100):
        """n learning rates over the whole schedule (for plots); does not touch the optimizer"""
        scale = self.end if self.units == "steps" else 1
        lrs = [self.get_lr_at_pos(i / n * scale) for i in range(n)]
        self.last_lr = 0
        return lrs

    def print_frames(self):
        for f in self.frames:
            print(f)
