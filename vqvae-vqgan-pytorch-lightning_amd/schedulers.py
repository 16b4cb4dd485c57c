"""Host-side scalar schedules with the call surface the reference uses from the (un-vendored, C++)
``scheduling_utils.schedulers_cpp`` package: ``Scheduler(start_step, stop_step, start_value, stop_value
[, th_step]).step(i)`` and ``.destroy()`` (call sites vqvae/model.py:175-200, :210, :222-224, :307).
PARITY UNPINNED: that package is not in the reference tree and not installed, so the formulas below are
the documented semantics (linear ramp, half-cosine, linear warm-up followed by half-cosine)."""
import math


class _Base:
    def __init__(self, start_step, stop_step, start_value, stop_value):
        self.start_step, self.stop_step = int(start_step), int(stop_step)
        self.start_value, self.stop_value = float(start_value), float(stop_value)

    def destroy(self):
        pass


class LinearScheduler(_Base):
    def step(self, i: int) -> float:
        if i <= self.start_step:
            return self.start_value
        if i >= self.stop_step:
            return self.stop_value
        t = (i - self.start_step) / (self.stop_step - self.start_step)
        return self.start_value + (self.stop_value - self.start_value) * t


class CosineScheduler(_Base):
    def step(self, i: int) -> float:
        if i <= self.start_step:
            return self.start_value
        if i >= self.stop_step:
            return self.stop_value
        t = (i - self.start_step) / (self.stop_step - self.start_step)
        return self.stop_value + 0.5 * (self.start_value - self.stop_value) * (1 + math.cos(math.pi * t))


class LinearCosineScheduler(_Base):
    """linear 0 -> start_value until th_step, then half-cosine start_value -> stop_value until stop_step"""

    def __init__(self, start_step, stop_step, start_value, stop_value, th_step):
        super().__init__(start_step, stop_step, start_value, stop_value)
        self.th_step = int(th_step)
        self._warm = LinearScheduler(start_step, th_step, 1e-20, start_value)
        self._cos = CosineScheduler(th_step, stop_step, start_value, stop_value)

    def step(self, i: int) -> float:
        return self._warm.step(i) if i < self.th_step else self._cos.step(i)
