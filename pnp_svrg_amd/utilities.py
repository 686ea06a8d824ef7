"""The step after the loop: reference Utilities.py:5-64 `display_results`.

Consumes the result dict every pnp_* loop returns (`z`, `time_per_iter`, `psnr_per_iter`, `gradient_time`,
`denoise_time`, `algo_name`) and produces the reference's artefacts: output image, PSNR-over-time plot, one metrics
line on stdout and `output.csv`.  Host-side matplotlib/csv only -- nothing here touches the device."""
import csv
import os

import numpy as np


def metrics_line(output_dict, reference_labels=True):
    """The line the reference prints (Utilities.py:51-53).  Its format string indexes the wrong arguments: the field
    labelled 'Change in PSNR' shows gradient_time and both time fields show denoise_time (BASELINE.md section 1 decodes
    the notebooks' logs through this).  reference_labels=True reproduces that byte for byte, False prints what the
    labels say."""
    psnr = output_dict['psnr_per_iter']
    args = (psnr[-1], psnr[-1] - psnr[0], output_dict['gradient_time'], output_dict['denoise_time'])
    if reference_labels:
        return 'Output PSNR: {0:3.1f}\tChange in PSNR: {2:3.2f}\tGradient Time: {3:3.2f}\tDenoising Time: {3:3.2f}'.format(*args)
    return 'Output PSNR: {0:3.1f}\tChange in PSNR: {1:3.2f}\tGradient Time: {2:3.2f}\tDenoising Time: {3:3.2f}'.format(*args)


def metrics_row(output_dict):
    """header, row of output.csv (Utilities.py:57-63; the CSV, unlike the printed line, is indexed correctly)."""
    psnr = output_dict['psnr_per_iter']
    return (['Output PSNR', 'Change in PSNR', 'Gradient Time', 'Denoising Time'],
            [np.around(psnr[-1], decimals=1), np.around(psnr[-1] - psnr[0], decimals=2),
             np.around(output_dict['gradient_time'], decimals=2), np.around(output_dict['denoise_time'], decimals=2)])


def display_results(problem, output_dict, save_results=False, save_dir='figures/', show_figs=False, *,
                    reference_labels=True):
    """reference Utilities.py:5-64.  Returns the PSNR axes like the reference (callers overlay further curves)."""
    import matplotlib.pyplot as plt
    base = None
    if save_results:
        prob_dir = getattr(problem, 'prob_dir', None)
        base = prob_dir + output_dict['algo_name'] + '/' if prob_dir else save_dir
        os.makedirs(base, exist_ok=True)

    out_fig = plt.figure(figsize=(6, 6))
    plt.imshow(np.asarray(output_dict['z']).reshape(problem.H, problem.W), cmap=getattr(problem, 'color_map', 'gray'),
               vmin=0, vmax=1)
    plt.title('Output Image')
    plt.xticks([])
    plt.yticks([])
    if save_results:
        out_fig.savefig(base + 'output.eps', transparent=True, bbox_inches='tight', pad_inches=0)
    if show_figs:
        plt.show()

    psnr_fig = plt.figure(figsize=(6, 6))
    psnr_ax = psnr_fig.add_subplot(1, 1, 1)
    t = np.cumsum(output_dict['time_per_iter'])
    psnr = np.asarray(output_dict['psnr_per_iter'])
    psnr_ax.plot(t, psnr, "b", linewidth=3, label=str(output_dict['algo_name']))
    psnr_ax.plot(t[::30], psnr[::30], "b*", markersize=10)
    psnr_ax.set(xlabel='time (s)', ylabel='PSNR (dB)')
    psnr_ax.legend()
    psnr_ax.grid()
    psnr_fig.tight_layout()
    if show_figs:
        plt.show()
    if save_results:
        psnr_fig.savefig(base + 'psnr_over_time.eps', transparent=True, bbox_inches='tight', pad_inches=0)

    print(metrics_line(output_dict, reference_labels))
    if save_results:
        header, row = metrics_row(output_dict)
        with open(base + 'output.csv', 'w') as f:
            w = csv.writer(f, delimiter=',')
            w.writerow(header)
            w.writerow(row)
    return psnr_ax
