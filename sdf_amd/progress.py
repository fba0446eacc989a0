"""Terminal progress widget with the reference's names (`sdf.progress.Bar`, `pretty_time`: reference
sdf/progress.py:4-9, 11-83) so that scripts importing `sdf.progress` keep working.  `generate` here does not
drive it -- the device meshes every batch of a call in one submission, there is nothing to tick -- but user
code that builds its own loops around `Bar` finds the same constructor and methods."""
import sys
import time


def pretty_time(seconds):
    """H:MM:SS of a duration in seconds"""
    s = int(round(seconds))
    return '%d:%02d:%02d' % (s // 3600, (s // 60) % 60, s % 60)


class Bar(object):
    def __init__(self, max_value=100, min_value=0, enabled=True):
        self.min_value, self.max_value, self.enabled = min_value, max_value, enabled
        self.value = min_value
        self.start_time = time.time()
        self._last = 0.0

    @property
    def percent_complete(self):
        span = self.max_value - self.min_value
        return 100.0 if span == 0 else 100.0 * (self.value - self.min_value) / span

    @property
    def elapsed_time(self):
        return time.time() - self.start_time

    @property
    def eta(self):
        p = self.percent_complete / 100.0
        return 0.0 if p <= 0 else self.elapsed_time * (1.0 - p) / p

    def increment(self, delta):
        self.update(self.value + delta)

    def update(self, value):
        self.value = value
        now = time.time()
        if self.enabled and now - self._last >= 0.1:       # at most ten redraws a second
            self._last = now
            self._draw('\r')

    def done(self):
        self.update(self.max_value)
        self.stop()

    def stop(self):
        if self.enabled:
            self._draw('\n')

    def render(self):
        filled = int(round(30 * self.percent_complete / 100.0))
        return '%3.0f%% (%g of %g) [%s%s] %s %s' % (
            self.percent_complete, self.value - self.min_value, self.max_value - self.min_value,
            '#' * filled, '-' * (30 - filled), pretty_time(self.elapsed_time), pretty_time(self.eta))

    def _draw(self, end):
        sys.stdout.write('  ' + self.render().ljust(78) + end)
        sys.stdout.flush()
