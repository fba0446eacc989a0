"""Console progress bar with the constructor / increment / done surface of reference
sdf/progress.py:11-82.  The device meshes all batches in one launch, so there is nothing to
tick per batch; the class exists for scripts that import it."""
import sys
import time


class Bar:
    def __init__(self, max_value=100, min_value=0, enabled=True):
        self.min_value = min_value
        self.max_value = max_value
        self.value = min_value
        self.start_time = time.time()
        self.enabled = enabled

    @property
    def percent_complete(self):
        span = self.max_value - self.min_value
        return 100.0 if span == 0 else 100.0 * (self.value - self.min_value) / span

    @property
    def elapsed_time(self):
        return time.time() - self.start_time

    def increment(self, delta):
        self.update(self.value + delta)

    def update(self, value):
        self.value = value
        if self.enabled:
            sys.stdout.write('  %3d%% \r' % int(self.percent_complete))
            sys.stdout.flush()

    def done(self):
        self.update(self.max_value)
        if self.enabled:
            sys.stdout.write('\n')
            sys.stdout.flush()
