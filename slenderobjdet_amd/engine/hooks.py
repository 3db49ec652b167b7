"""Training hooks (detectron2.engine.hooks subset the reference registers, slender_det/engine/defaults.py:89-134, and its
own EvalHookRefine, slender_det/engine/hooks.py:18-43)."""
import time


class HookBase:
    trainer = None

    def before_train(self):
        pass

    def after_train(self):
        pass

    def before_step(self):
        pass

    def after_step(self):
        pass


class IterationTimer(HookBase):
    def __init__(self, warmup_iter=3):
        self._warmup_iter = warmup_iter
        self._start = None
        self.total = 0.0
        self.count = 0

    def before_step(self):
        self._start = time.perf_counter()

    def after_step(self):
        it = self.trainer.iter - self.trainer.start_iter + 1
        if it > self._warmup_iter:
            self.total += time.perf_counter() - self._start
            self.count += 1


class LRScheduler(HookBase):
    def __init__(self, optimizer, scheduler):
        self._optimizer, self._scheduler = optimizer, scheduler

    def after_step(self):
        self.trainer.storage["lr"] = self._optimizer.param_groups[0]["lr"]
        self._scheduler.step()


class PeriodicWriter(HookBase):
    def __init__(self, period=20, log=print):
        self._period, self._log = period, log

    def after_step(self):
        t = self.trainer
        if (t.iter + 1) % self._period == 0 or t.iter == t.max_iter - 1:
            losses = " ".join(f"{k}: {v:.4f}" for k, v in t.storage.get("losses", {}).items())
            self._log(f"iter: {t.iter + 1}  total_loss: {t.storage.get('total_loss', float('nan')):.4f}  {losses}  lr: {t.storage.get('lr', 0):.6f}")


class PeriodicCheckpointer(HookBase):
    def __init__(self, checkpointer, period):
        self.checkpointer, self.period = checkpointer, int(period)

    def after_step(self):
        it = self.trainer.iter + 1
        if self.period > 0 and it % self.period == 0:
            self.checkpointer.save(f"model_{it - 1:07d}", iteration=it - 1)


class EvalHook(HookBase):
    def __init__(self, eval_period, eval_function):
        self._period, self._func = eval_period, eval_function

    def after_step(self):
        nxt = self.trainer.iter + 1
        if self._period > 0 and nxt % self._period == 0 and nxt != self.trainer.max_iter:
            self._do_eval()

    def after_train(self):
        if self._period > 0 and self.trainer.iter + 1 >= self.trainer.max_iter:
            self._do_eval()

    def _do_eval(self):
        results = self._func()
        if results:
            self.trainer.storage.setdefault("eval", []).append(results)


class EvalHookRefine(EvalHook):
    """slender_det/engine/hooks.py:18-43: like EvalHook, keeping only float-convertible metrics."""

    def _do_eval(self):
        results = self._func()
        if results:
            flat = {}
            for k, v in results.items():
                try:
                    flat[k] = float(v)
                except (TypeError, ValueError):
                    continue
            self.trainer.storage.setdefault("eval", []).append(flat)
