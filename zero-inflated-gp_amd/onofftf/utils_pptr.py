"""`preprocessing` -- onofftf/utils_pptr.py:4-123: optional time-window filter and min/range scaling of the precipitation
data set (pandas frames traindf/testdf with columns lat, lon, ndatehour next to the Xtrain/Xtest arrays whose columns
are in that order), and the heuristic initial kernel parameters derived from it.  Pure host-side data preparation."""
import numpy as np

_KEYS = ('traindf', 'testdf', 'Xtrain', 'Ytrain', 'Xtest', 'Ytest')
_COLS = ('lat', 'lon', 'ndatehour')


class preprocessing:
    def __init__(self, data):
        self.data = {'raw': {k: data[k] for k in _KEYS}}
        self.filter_flg = False
        self.scale_flg = False

    def filter_time(self, min_idx=0, max_idx=np.inf):
        """keep rows whose ndatehour lies in [min_idx, max_idx]  (:16-27)"""
        self.filter_flg = True
        raw = self.data['raw']
        keep = {s: np.asarray((raw[s + 'df'].ndatehour >= min_idx) & (raw[s + 'df'].ndatehour <= max_idx)) for s in ('train', 'test')}
        self.data['filt'] = {'traindf': raw['traindf'][keep['train']], 'testdf': raw['testdf'][keep['test']],
                             'Xtrain': raw['Xtrain'][keep['train']], 'Ytrain': raw['Ytrain'][keep['train']],
                             'Xtest': raw['Xtest'][keep['test']], 'Ytest': raw['Ytest'][keep['test']]}

    def _span(self, col):
        d = self.data['scaled']
        lo = min(d['traindf'][col].min(), d['testdf'][col].min())
        hi = max(d['traindf'][col].max(), d['testdf'][col].max())
        return {'min': lo, 'range': hi - lo}

    def scale(self, scale_loc=False, scale_time=False):
        """(x - min) / range over train+test for lat/lon and/or ndatehour, in place on the selected stage  (:29-87)"""
        self.scale_flg_loc, self.scale_flg_time = bool(scale_loc), bool(scale_time)
        if scale_loc or scale_time:
            self.scale_flg = True
        self.data['scaled'] = self.data['filt'] if self.filter_flg else self.data['raw']
        if scale_loc:
            self.scale_param = {'lat': self._span('lat'), 'lon': self._span('lon')}
        if scale_time:
            self.scale_param.update({'ndatehour': self._span('ndatehour')})      # as the reference: needs scale_loc too (:56)
        if not self.scale_flg:
            return
        todo = ([0, 1] if scale_loc else []) + ([2] if scale_time else [])
        for key in ('Xtrain', 'Xtest', 'traindf', 'testdf'):
            ds = self.data['scaled'][key]
            for c in todo:
                sp = self.scale_param[_COLS[c]]
                if isinstance(ds, np.ndarray):
                    ds[:, c] = (ds[:, c] - sp['min']) / sp['range']
                else:
                    ds[_COLS[c]] = (ds[_COLS[c]] - sp['min']) / sp['range']
            self.data['scaled'][key] = ds

    def _stage(self):
        return 'scaled' if self.scale_flg else ('filt' if self.filter_flg else 'raw')

    @property
    def model_data(self):
        return self.data[self._stage()]

    @property
    def shape(self):
        d = self.data[self._stage()]
        return d['Xtrain'].shape, d['Xtest'].shape

    @property
    def kernel_params(self):
        """(variance, [l_lat, l_lon, l_time]): max target, lengthscale 3 in raw units = round(3 / range, 4) once scaled  (:100-123)"""
        variance = np.max(self.model_data['Ytrain'])
        ell = [3., 3., 3.]
        if self.scale_flg and self.scale_flg_loc:
            ell[0] = round(3. / self.scale_param['lat']['range'], 4)
            ell[1] = round(3. / self.scale_param['lon']['range'], 4)
        if self.scale_flg and self.scale_flg_time:
            ell[2] = round(3. / self.scale_param['ndatehour']['range'], 4)
        return (variance, ell)
