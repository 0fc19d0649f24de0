"""Minimal stand-in for util/html.py:6-49 (dominate is not a dependency here): test.py only uses the
page object for its directory layout -- `<web_dir>/images/` -- and never calls .save()."""
import os


class HTML:
    def __init__(self, web_dir, title, reflesh=0):
        self.title = title
        self.web_dir = web_dir
        self.img_dir = os.path.join(self.web_dir, 'images')
        os.makedirs(self.img_dir, exist_ok=True)
        self.rows = []

    def get_image_dir(self):
        return self.img_dir

    def add_header(self, text):
        self.rows.append('<h3>%s</h3>' % text)

    def add_images(self, ims, txts, links, width=400):
        cells = ''.join('<td><a href="%s"><img style="width:%dpx" src="%s"></a><br><p>%s</p></td>'
                        % (os.path.join('images', l), width, os.path.join('images', im), t) for im, t, l in zip(ims, txts, links))
        self.rows.append('<table border="1" style="table-layout: fixed;"><tr>%s</tr></table>' % cells)

    def save(self):
        with open(os.path.join(self.web_dir, 'index.html'), 'wt') as f:
            f.write('<html><head><title>%s</title></head><body>%s</body></html>' % (self.title, '\n'.join(self.rows)))
