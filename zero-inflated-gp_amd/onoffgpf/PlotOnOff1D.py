"""`PlotOnOff1D(m)` -- onoffgpf/PlotOnOff1D.py:8-167: the 4x4 summary figure of a fitted 1-D zero-inflated GP (panels a-h of
the paper's toy figure).  `panel_data(m)` computes everything the figure shows (predictions through the engine, kernel
images through zigp_rbf_K); `PlotOnOff1D(m, fname)` draws it.  matplotlib is imported lazily: the model code never needs it."""
import numpy as np


def panel_data(m):
    """Curves, 1.5-sigma / 2-sigma bands and kernel images, all on the training inputs  (PlotOnOff1D.py:12-41,56-57,67-68,78-79)."""
    X, Y = m.Xtrain.value, m.Ytrain.value
    gfmean, gfvar, _, fmean, fvar, gmean, gvar, pgmean, pgvar = [np.asarray(a).reshape(-1) for a in m.predict_onoffgp(X)]
    Kf, Kg = m.kernf.compute_K_symm(X), m.kerng.compute_K_symm(X)
    Kpg = np.outer(pgmean, pgmean)                              # Phi(g) Phi(g)^T                       (:28)
    noise_sd = np.sqrt(float(np.asarray(m.likelihood.variance.value).reshape(-1)[0]))
    mix_sd = np.sqrt(fvar) * pgmean + np.sqrt(pgvar) * (1 - pgmean)   # spread of f|g used by the reference  (:56,78)
    return dict(x=X.reshape(-1), y=Y.reshape(-1), gfmean=gfmean, gfvar=gfvar, fmean=fmean, fvar=fvar, gmean=gmean, gvar=gvar,
                pgmean=pgmean, pgvar=pgvar, Kf=Kf, Kg=Kg, Kpg=Kpg, Kfg=Kpg * Kf,
                y_band=(gfmean - 1.5 * (mix_sd + noise_sd), gfmean + 1.5 * (mix_sd + noise_sd)),
                f_band=(fmean - 1.5 * np.sqrt(fvar), fmean + 1.5 * np.sqrt(fvar)),
                fg_band=(gfmean - 1.5 * mix_sd, gfmean + 1.5 * mix_sd),
                pg_band=(pgmean - 2 * np.sqrt(pgvar), pgmean + 2 * np.sqrt(pgvar)),
                g_band=(gmean - 2 * np.sqrt(gvar), gmean + 2 * np.sqrt(gvar)),
                Zf=m.Zf.value.reshape(-1), u_fm=m.u_fm.value.reshape(-1), Zg=m.Zg.value.reshape(-1), u_gm=m.u_gm.value.reshape(-1))


def PlotOnOff1D(m, fname='plots/toy.png', show=False):
    import os
    import matplotlib as mpl
    import matplotlib.pyplot as plt
    from matplotlib import gridspec, ticker
    d = panel_data(m)
    ORANGE, GREEN, NAVY, STEEL, GREY = '#ff7707', '#008b62', '#003366', '#6684a3', '#333333'
    with mpl.rc_context({'figure.figsize': (11.0, 10.0), 'font.size': 20}):
        fig = plt.figure()
        grid = gridspec.GridSpec(4, 4)
        curves = [fig.add_subplot(grid[r, 0:-1]) for r in range(4)]
        images = [fig.add_subplot(grid[r, -1]) for r in range(4)]
        x = d['x']

        def band(ax, key, color, alpha):
            ax.fill_between(x, d[key][0], d[key][1], facecolor=color, alpha=alpha)

        a = curves[0]                                                       # (a) y
        a.plot(x, d['gfmean'], '-', color=ORANGE)
        band(a, 'y_band', ORANGE, 0.5)
        a.scatter(x, d['y'], s=8, color='black', alpha=0.7)
        a.set_title('(a) Predictive function \n' + r'$\mathbf{y}$', fontsize=18)
        a = curves[1]                                                       # (c) f and f|g with the inducing values
        a.plot(x, d['fmean'], '-', color=GREEN, label=r'$f$')
        band(a, 'f_band', GREEN, 0.5)
        a.plot(d['Zf'], d['u_fm'], marker='o', linestyle='None', markeredgecolor='None', markerfacecolor=GREEN, alpha=0.7)
        a.plot(x, d['gfmean'], '-', color=ORANGE, label=r'$f|g$')
        band(a, 'fg_band', ORANGE, 0.5)
        a.set_title('(c) Sparse latent function \n' + r'$\mathbf{f}|\mathbf{g}$', fontsize=18)
        a.legend(loc='lower right', ncol=1, fontsize=18)
        a = curves[2]                                                       # (e) Phi(g)
        a.plot(x, d['pgmean'], '-', color=NAVY)
        band(a, 'pg_band', STEEL, 0.7)
        a.axhline(y=0.5, linestyle='--', color=GREY)
        a.set_title('(e) Probit support function \n' + r'$\Phi(\mathbf{g})$', fontsize=18)
        a = curves[3]                                                       # (g) g with the inducing values
        a.plot(x, d['gmean'], '-', color=NAVY)
        a.plot(d['Zg'], d['u_gm'], marker='o', linestyle='None', markeredgecolor='None', markerfacecolor=NAVY, alpha=0.8)
        band(a, 'g_band', STEEL, 0.7)
        a.axhline(y=0.0, linestyle='--', color=GREY)
        a.set_title('(g) Latent function \n' + r'$\mathbf{g}$', fontsize=18)
        for i, a in enumerate(curves):
            a.set_xlim(0, 10)
            if i < 3:
                a.set_xticks([])
            if i < 2:
                a.set_yticks([-1, 0, 1])
        titles = (('Kfg', '(b) Sparse kernel \n' + r'$\Phi(\mathbf{g}) \Phi(\mathbf{g})^T \circ K_f$'), ('Kf', '(d) Latent kernel \n' + r'$K_f$'),
                  ('Kpg', '(f) Probit kernel \n' + r'$\Phi(\mathbf{g}) \Phi(\mathbf{g})^T$'), ('Kg', '(h) Latent kernel \n' + r'$K_g$'))
        for a, (key, title) in zip(images, titles):
            im = a.imshow(d[key], cmap='viridis')
            cb = fig.colorbar(im, ax=a, fraction=0.046, pad=0.03, extend='max')
            cb.locator = ticker.MaxNLocator(nbins=4)
            cb.update_ticks()
            a.set_title(title, fontsize=18)
            a.set_xticks([])
            a.set_yticks([])
        fig.tight_layout()
        fig.subplots_adjust(hspace=0.5, wspace=0.1)
        if fname:
            os.makedirs(os.path.dirname(fname) or '.', exist_ok=True)
            fig.savefig(fname)
        if show:
            plt.show()
        plt.close(fig)
    return d
