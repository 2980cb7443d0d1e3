"""ITU-R tapped-delay-line channel realisations (host, numpy) -- the inputs of the hot path.

Mirrors ``python/channel_model/itur_channels.py:14-94`` (``gen_chan``) and
``python/channel_model/rayleigh_fading.py:52-102`` (GMEDS_1 sum of sinusoids), vectorised,
including the reference's quirks: the sinc interpolation grid is
``linspace(-(n-1)/2, (n+1)/2, n)`` (step 1.05) and, with one frame, every path has the
deterministic magnitude sqrt(PDP) and a random phase.  Drawing order of the random phases is
the reference's (waveform, oscillator, real-then-imaginary), so the same numpy seed gives the
same channels.
"""
import numpy as np

#: itur_channels.py:14-30 -- relative delay [s], average power [dB]
CHANNEL_ITUR = {
    "vehicularA": ([0, 310e-9, 710e-9, 1090e-9, 1730e-9, 2510e-9], [0, -1, -9, -10, -15, -20]),
    "vehicularB": ([0, 300e-9, 8900e-9, 12900e-9, 17100e-9, 20000e-9], [-2.5, 0, -12.8, -10, -25.2, -16]),
    "outdoor-indoorA": ([0, 110e-9, 190e-9, 410e-9], [0, -9.7, -19.2, -22.8]),
    "outdoor-indoorB": ([0, 200e-9, 800e-9, 1200e-9, 2300e-9, 3700e-9], [0, -.9, -4.9, -8, -7.8, -23.9]),
}


def rayleigh_fading_gmeds_1(no_oscillators, doppler_freq, no_channels, sampling_freq, no_waveforms,
                            rng=None):
    """[no_channels, no_waveforms] complex Rayleigh waveforms (rayleigh_fading.py:52-102)."""
    rng = np.random if rng is None else rng
    t = np.arange(0, no_channels, 1) / sampling_freq                       # [T]
    w = np.arange(no_waveforms)[:, None]
    o = np.arange(no_oscillators)[None, :]
    rot = (np.pi / (4 * no_oscillators)) * (w / (no_waveforms + 2))
    arr = (np.pi / (2 * no_oscillators)) * (o + .5)
    phases = np.pi * rng.randn(no_waveforms, no_oscillators, 2)            # (real, imag) per oscillator
    f_re = doppler_freq * np.cos(arr + rot)
    f_im = doppler_freq * np.cos(arr - rot)
    re = np.cos(2 * np.pi * f_re[..., None] * t + phases[..., 0:1]).sum(axis=1)   # [W, T]
    im = np.cos(2 * np.pi * f_im[..., None] * t + phases[..., 1:2]).sum(axis=1)
    return (np.sqrt(2 / no_oscillators) * (re + 1j * im)).T


def gen_chan(standard, no_samples, doppler_freq, sampling_rate, frame_duration, no_frames, rng=None):
    """[no_samples, no_frames] complex tapped-delay-line model (itur_channels.py:33-94)."""
    delays, powers = CHANNEL_ITUR[standard]
    delays, powers = np.asarray(delays), np.asarray(powers, dtype=np.float64)
    waves = rayleigh_fading_gmeds_1(21, doppler_freq, no_frames, 1 / frame_duration, len(delays), rng)
    wave_power = np.mean(np.abs(waves) ** 2, axis=0)                       # adjust_power, 68-75
    coef = (np.sqrt(10 ** (powers / 10) / wave_power) * waves).T           # [paths, frames]
    axis = np.linspace(-(no_samples - 1) / 2, (no_samples + 1) / 2, no_samples)
    sinc_mat = np.sinc(delays[None, :] * sampling_rate - axis[:, None])   # [samples, paths]
    return sinc_mat @ coef


def gen_channel_file(standard="vehicularA", no_channels=250, no_samples=21, carrier_frequency=2e9,
                     sample_period=200e-9, velocity=100 / 3.6, no_symbols=16, dft_length=256,
                     rng=None):
    """The ``gen_chan`` mode of ``python/wofdm_optimization.py:63-86``: [no_samples, no_channels]
    (column = realisation), ready for ``np.save(channels/<standard>.npy)``."""
    speed_of_light = 299792458.0
    doppler = (velocity / speed_of_light) * carrier_frequency
    frame_duration = no_symbols * dft_length * sample_period
    out = np.zeros((no_samples, no_channels), dtype=np.complex128)
    for i in range(no_channels):
        out[:, i] = gen_chan(standard, no_samples, doppler, 1 / sample_period, frame_duration, 1, rng)[:, 0]
    return out
