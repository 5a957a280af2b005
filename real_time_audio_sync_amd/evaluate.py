"""Alignment-accuracy metric of the reference's harness (tests.py:29-137, class ``test_simple``) and
its ground-truth CSV format (``time_seconds,beat_index[,label]``, Songs/<piece>/<recording>.csv).

Host logic (a few hundred path points per recording pair); kept bug-for-bug so numbers are
comparable with the notebooks (livenote_v2.ipynb:336-343):
  * a path point is skipped when either interpolated beat is None *or falsy* (``if l_beat and
    r_beat`` -- a beat of exactly 0.0 is dropped, tests.py:71);
  * ``get_time`` indexes the *live* ground truth with the integer part of the beat for both the
    reference and the live beat (tests.py:130-137)."""
import csv

FRAME_SECONDS = 2048 / 22050.  # tests.py:114


def read_ground_truth(csv_path):
    """-> (times [s], beats [int]) from a Songs/**.csv file (tests.py:45-57)."""
    times, beats = [], []
    with open(csv_path) as fh:
        for row in csv.reader(fh):
            times.append(float(row[0]))
            beats.append(int(row[1]))
    return times, beats


def get_beat(sample, gt_times, gt_beats, frame_seconds=FRAME_SECONDS):
    """Fractional beat of chroma frame ``sample``: linear interpolation between annotated beat times,
    the stretch before the first annotation being interpolated from time zero; None past the last
    annotation.  Same arithmetic as tests.py:112-128: beat_i - (t_i - t) / (t_i - t_{i-1})."""
    t = sample * frame_seconds
    seg_start = 0.0  # the first segment starts at time zero, later ones at the previous annotation
    for i, (seg_end, beat) in enumerate(zip(gt_times, gt_beats)):
        inside = (t <= seg_end) if i == 0 else (seg_start <= t <= seg_end)
        if inside:
            if i == 0 and seg_end == 0:
                return beat - 0
            return beat - float(seg_end - t) / (seg_end - seg_start)
        seg_start = seg_end
    return None


class AlignmentError(object):
    """``AlignmentError(ref_csv, live_csv, path).get_error()`` == ``test_simple(ref_wav, live_wav,
    path).get_error()`` of the reference (which derives the CSV names from the WAV names)."""

    THRESHOLDS = (1, 3, 5, 10)

    def __init__(self, ref_csv, live_csv, path):
        self.ref_gt_times, self.ref_gt_beats = read_ground_truth(ref_csv)
        self.live_gt_times, self.live_gt_beats = read_ground_truth(live_csv)
        self.path = path

    def get_time(self, beat):
        """Seconds for a fractional beat, read off the *live* annotation with the beat's integer part used
        directly as the list index (the reference's convention, tests.py:130-134)."""
        times = self.live_gt_times
        whole = int(beat)
        seconds = times[whole]
        if whole + 1 < len(times):
            seconds += (beat % 1) * (times[whole + 1] - times[whole])
        return seconds

    def get_secs_off(self, ref_beat, live_beat):
        return abs(self.get_time(ref_beat) - self.get_time(live_beat))

    def stats(self):
        """dict: count, squared beat error sum, % of points off by more than 1/3/5/10 beats / seconds."""
        off_beats = [0, 0, 0, 0]
        off_secs = [0, 0, 0, 0]
        error = 0.0
        count = 0
        for (l, r) in self.path:
            l_beat = get_beat(l, self.live_gt_times, self.live_gt_beats)
            r_beat = get_beat(r, self.ref_gt_times, self.ref_gt_beats)
            if l_beat and r_beat:
                diff = abs(l_beat - r_beat)
                error += diff ** 2
                seconds_off = self.get_secs_off(r_beat, l_beat)
                for n, thr in enumerate(self.THRESHOLDS):
                    if diff > thr:
                        off_beats[n] += 1
                    if seconds_off > thr:
                        off_secs[n] += 1
                count += 1
        if count == 0:
            raise ZeroDivisionError("no path point falls inside the annotated range")  # as the reference
        return dict(count=count, squared_beat_error=error,
                    pct_off_beats={t: float(v) / count * 100 for t, v in zip(self.THRESHOLDS, off_beats)},
                    pct_off_seconds={t: float(v) / count * 100 for t, v in zip(self.THRESHOLDS, off_secs)})

    def get_error(self, verbose=True):
        s = self.stats()
        if verbose:
            for t in self.THRESHOLDS:
                print("Percent incorrect (within %d beat%s): %s %%" % (t, "" if t == 1 else "s", s["pct_off_beats"][t]))
            for t in self.THRESHOLDS:
                print("Percent incorrect (within %d second%s): %s %%" % (t, "" if t == 1 else "s", s["pct_off_seconds"][t]))
        return s["pct_off_seconds"][3]  # tests.py:109
